import os, sys, time, subprocess
sys.path.insert(0, os.getcwd())
code = '''
import os, sys, time
sys.path.insert(0, os.getcwd())
import ray_tracer_amd as rt
rt.lib()
for tol in (False, True):
    best = 9
    for _ in range(3):
        t = time.time(); h = rt.HostScene.load_obj("/tmp/dragon_standin.obj", "dragon", no_bvh=True, tolerant=tol); best = min(best, time.time() - t)
    print(os.environ.get("OMP_NUM_THREADS"), "threads tolerant" if tol else "threads strict", round(best, 3), flush=True)
'''
import ray_tracer_amd as rt
rt.HostScene.standin("dragon", seed=0, obj_path_out="/tmp/dragon_standin.obj", no_bvh=True)
for n in (1, 4, 16):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS=str(n), RTX_TIMING="1"))
