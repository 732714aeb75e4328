#!/usr/bin/env python3
"""Census of one 4K frame with the library named by RTX_LIB (a -DRTX_EXPERIMENT_STALE build re-purposes two counters:
leaf_visits = pair steps of nodes that came off the stack, box_hits = those of them that found every entered child
beyond the best hit).  usage: tools/stale_probe.py <scene> <family>"""
import os, sys
sys.path.insert(0, os.getcwd())
import ray_tracer_amd as rt
scene, family = sys.argv[1], sys.argv[2]
host = rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family)
dev = rt.Scene(host)
f = dev.render(3840, 2160, traversal=rt.TRAVERSAL_PRUNED, want_stats=True, want_hit=False)
d = f.stats.as_dict()
pairs = f.stats.lane_steps[0]
print(scene, family, d)
print(f"pair steps {pairs}  of nodes off the stack {d['leaf_visits']} ({d['leaf_visits']/pairs:.3f})  of those, nothing but children beyond the best hit {d['box_hits']} ({d['box_hits']/pairs:.3f} of all pair steps)")
