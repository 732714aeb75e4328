"""The drop-in carried out inside the reference's own program: oracle/_ref/ref_dropin is the reference's Scene
factory, loader and BVH constructor (its headers, compiled in the build container), the binding code a maintainer
would add (scene arrays handed to rtx_scene_create), one rtx_render() call in place of the thread fan-out, and the reference's own per-pixel loop run
beside it on host threads; it exits 0 when every primary triangle id and every colour agree."""
import os
import shutil
import subprocess

import pytest

from oracle_binding import ORACLE_DIR

pytestmark = pytest.mark.gpu
DROPIN = os.path.join(ORACLE_DIR, "_ref", "ref_dropin")


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref is only built where /root/reference exists")
@pytest.mark.parametrize("mode,own_tree", [("reference", 0), ("pruned", 0), ("pruned", 1)])
def test_reference_main_with_the_binding_renders_the_reference_frame(teapot_obj, tmp_path, mode, own_tree):
    os.makedirs(tmp_path / "scenes")
    shutil.copyfile(teapot_obj, tmp_path / "scenes" / "teapot.obj")      # where Scene::teapot() looks (scene.hpp:262)
    out = subprocess.run([DROPIN, str(tmp_path), "640", "400", mode, str(own_tree)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pixels 256000 id_mismatches 0 rgb_mismatches 0" in out.stdout
    assert "12541 nodes, identical to BVH::nodes / BVH::indices" in out.stdout
