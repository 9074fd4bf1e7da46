"""The queued kernel edits under experiments/ are patches against sunscreen_amd/csrc/: each must still apply to the tree as it
is (tools/build_patched_variant.sh builds them as variant libraries).  A patch that has been merged -- or overtaken by another
edit -- fails here on purpose: delete it (or refresh it) in the same commit."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = sorted(glob.glob(os.path.join(ROOT, "experiments", "*", "*.patch")))


@pytest.mark.parametrize("patch", PATCHES, ids=[os.path.basename(p) for p in PATCHES])
def test_queued_patch_applies_to_the_tree(patch, tmp_path):
    dst = tmp_path / "sunscreen_amd" / "csrc"
    shutil.copytree(os.path.join(ROOT, "sunscreen_amd", "csrc"), dst)
    out = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "FAILED" not in out.stdout and "fuzz" not in out.stdout, out.stdout  # exact context: the hunks were cut from this tree
