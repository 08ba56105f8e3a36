"""C-ABI library: builds, loads and exports every symbol include/gumbi_hip.h declares.  No compute
call is made here (no GPU in the build container); the product path must refuse to run without
a device instead of falling back to the CPU."""
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from gumbi_amd import build, engine

    build.build_library()
    return engine.load_library()


def test_header_symbols_are_all_exported(lib):
    from gumbi_amd import engine

    header = (ROOT / "include" / "gumbi_hip.h").read_text()
    declared = set(re.findall(r"\b(gmb_[a-z_0-9]+)\s*\(", header))
    assert declared, "no prototypes parsed from the header"
    assert declared == set(engine.exported_symbols()), declared ^ set(engine.exported_symbols())
    for name in declared:
        assert hasattr(lib, name), f"{name} missing from libgumbi_hip.so"
    header = (ROOT / "include" / "gumbi_hip.h").read_text()
    declared = int(re.search(r"#define\s+GMB_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.gmb_abi_version() == declared == engine.ABI_VERSION


def test_struct_layouts_match_header():
    from gumbi_amd import engine

    # gmb_kernel_spec: 3 + 16 + 1 + 8 + 1 + 4 + 4 + 4 int32 (= 41), 4 bytes of padding, then one double
    assert engine.C.sizeof(engine._Spec) == 41 * 4 + 4 + 8
    # gmb_timings: every member is 8 bytes (double / int64_t) -- count them in the header, and compare names
    header = (ROOT / "include" / "gumbi_hip.h").read_text()
    body = re.search(r"typedef struct gmb_timings \{(.*?)\} gmb_timings;", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:double|int64_t)\s+([a-z_0-9]+)\s*;", body)
    assert fields == [n for n, _ in engine.Timings._fields_]
    assert engine.C.sizeof(engine.Timings) == 8 * len(fields) == 31 * 8
    # gmb_comm {int32 rank, world; void* ctx; fn* all_gather} and gmb_dist_step {9 x int32, pad, int64}
    assert engine.C.sizeof(engine.GmbComm) == 24 and engine.C.sizeof(engine.DistStep) == 48
    spec = engine.KernelSpec(D=6, idx_cont=[0, 1, 2], idx_lin=[1], coreg=[(3, 4)], out_col=5, n_out=2)
    cs = spec.to_c()
    assert (cs.n_cont, cs.n_lin, cs.n_coreg, cs.out_col, cs.n_out, cs.hetero_noise) == (3, 1, 1, 5, 2, 1)
    assert cs.jitter == 1e-6 and list(cs.idx_cont)[:3] == [0, 1, 2]
    assert spec.theta_size() == 3 + 2 + 2 + 12 + 6 + 6
    lib = engine.load_library()
    assert lib.gmb_theta_size(engine.C.byref(cs)) == spec.theta_size()
    with pytest.raises(ValueError):
        engine.KernelSpec(D=2, idx_cont=[0], kind="RatQuad").to_c()
    with pytest.raises(ValueError):
        engine.KernelSpec(D=20, idx_cont=list(range(17))).to_c()


def test_no_cpu_fallback_without_device(lib):
    from gumbi_amd import engine

    if lib.gmb_device_count() > 0:
        pytest.skip("a GPU is visible here; the refusal path is exercised on CPU-only machines")
    assert lib.gmb_device_count() == engine.GMB_ENODEVICE
    with pytest.raises(engine.GumbiHipError, match="no CPU fallback"):
        engine.Engine()
    with pytest.raises(engine.GumbiHipError):
        engine.ls_limits(np.zeros((4, 2)), ard=False)


def test_product_package_never_imports_oracle():
    for path in (ROOT / "gumbi_amd").rglob("*.py"):
        text = path.read_text()
        assert "oracle" not in text.replace("no CPU fallback", ""), f"{path} mentions the oracle"
    for path in (ROOT / "gumbi_amd" / "csrc").glob("*"):
        if path.suffix in (".hip", ".hpp"):
            assert "oracle/" not in path.read_text().replace("see oracle/gp_oracle.py square_dist", "")
