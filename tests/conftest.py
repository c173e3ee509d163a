import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build libvdl.so (hipcc cross-compiles on CPU) and the oracle once per session."""
    import __graft_entry__ as g

    if not os.path.exists(os.path.join(ROOT, "mplan2vdl_amd", "lib", "libvdl.so")) or \
            not os.path.exists(os.path.join(ROOT, "oracle", "libvdl_oracle.so")):
        g.build()
    return True


@pytest.fixture(scope="session", autouse=True)
def _parse_trap():
    """Evidence trap for the intermittent failure of DESIGN.md section 8 (twice in this build's history a program TEXT was read with one
    byte changed -- "t.g" as "t,g" by the engine's parser, "MaterializeCompact" cut after "Mate" by the oracle's -- and was fine on the
    next attempt).  Test infrastructure, not product code: every Engine.parse of the suite goes through this wrapper; a refusal is
    re-examined on the spot -- are the bytes handed over still the bytes of the str? does the same text parse on a second attempt? --
    and, if it was transient, reported as INTERMITTENT with the address of the buffer and the process's memory map beside it
    (gpurun_out/mismatch/), so that a hit can be laid against tools/heapguard's log of pinned ranges and quarantined blocks."""
    import mplan2vdl_amd.engine as eng_mod

    plain = eng_mod.Engine.parse

    def parse(self, vdl_text):
        try:
            return plain(self, vdl_text)
        except eng_mod.VdlError as first:
            if first.code != eng_mod._lib.VDL_ERR_PARSE or not isinstance(vdl_text, str):
                raise
            try:
                again = plain(self, vdl_text)
            except eng_mod.VdlError:
                raise first
            again.close()
            root = os.environ.get("GRAFT_REPO_ROOT") or ROOT
            d = os.path.join(root, "gpurun_out", "mismatch")
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "intermittent_parse_%d.txt" % os.getpid())
            with open(path, "a") as f:
                f.write("first attempt: %s\nstr object at 0x%x (%d chars)\n---- text\n%s\n---- /proc/self/maps\n%s\n" % (first, id(vdl_text), len(vdl_text), vdl_text, open("/proc/self/maps").read()))
            raise AssertionError("INTERMITTENT PARSE FAILURE: %s on the first attempt, accepted on the second (same str object at 0x%x); details in %s" % (first, id(vdl_text), path))

    eng_mod.Engine.parse = parse
    yield
    eng_mod.Engine.parse = plain


def golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return f.read()


@pytest.fixture(scope="session")
def q6_text():
    return golden("q6.vdl")


@pytest.fixture(scope="session")
def q1_text():
    return golden("q1.vdl")
