import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vit-deep-radiomics_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no libvdr.so (built artefacts are git-ignored): build it once, exactly as
    # __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU, ~1.5 min with -j8)
    if not os.path.exists(os.path.join(PKG, "vdr", "libvdr.so")):
        import subprocess
        r = subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j8"], capture_output=True, text=True)
        if r.returncode != 0:
            raise pytest.UsageError("libvdr.so build failed:\n" + r.stderr[-4000:])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
