import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    # The native artefacts are git-ignored (they travel with gpurun snapshots): on a fresh checkout build them once
    # (hipcc cross-compiles gfx950 without a GPU; gcc builds the C oracle).  Never silently skipped: a missing
    # toolchain makes the build -- and therefore the suite -- fail loudly.
    need = [os.path.join(ROOT, "dpr_scale_amd", "libdprhot.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.isfile(p) for p in need):
        import __graft_entry__

        __graft_entry__.build()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


def golden_names(prefix=""):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix))


@pytest.fixture(scope="session")
def golden():
    return load_golden
