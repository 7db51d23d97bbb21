"""CPU-only checks of the drop-in boundary: libmor_hip.so loads, exports every symbol that
include/mor_hip.h declares, agrees with the oracle on the params layout, and fails loudly
(no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re

import pytest

from dynamicslamtool_amd import MorParams, engine, kitti_params
from oracle import oracle as oracle_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "mor_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mor_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = engine.lib()
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "libmor_hip.so does not export %s" % name
    assert sorted(engine.EXPORTS) == declared, "engine.EXPORTS out of sync with include/mor_hip.h"


def test_params_layout_matches_oracle_and_header():
    assert engine.lib().mor_sizeof_params() == C.sizeof(MorParams) == oracle_mod.lib().oracle_sizeof_params()


def test_no_cpu_fallback():
    if engine.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(engine.MorError, match="no HIP device|no CPU fallback"):
        engine.MorBatch(kitti_params(), 1, 1024)


def test_product_never_references_oracle():
    """The product path must not import, link or call the oracle."""
    pkg = os.path.join(ROOT, "dynamicslamtool_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_" not in src and "libmor_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f
    import subprocess
    out = subprocess.run(["ldd", engine.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_no_autobuild_under_a_profiler():
    """engine.lib() rebuilds a stale library by itself — but never when a profiler's preload is in the environment (rocprofv3's tool library initialises the GPU in
    every process that inherits it, and hipcc is a chain of exec hops; a GPU-initialised process replacing its program takes the box down on this pool)."""
    assert not engine._profiler_preload({"PATH": "/usr/bin", "HOME": "/root"})
    for env in ({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"}, {"ROCP_TOOL_LIBRARIES": "x.so"}, {"ROCPROFILER_LIBRARY_CTOR": "1"}, {"ROCPROF_OUTPUT_PATH": "/tmp"}, {"HSA_TOOLS_LIB": "libx.so"}):
        assert engine._profiler_preload(env), env
    for script in ("profiles/collect.sh", "exp/pmc_atomics.sh", "exp/pmc_mem.sh", "exp/pmc_cg_phases.sh"):
        lines = [l.split("   #")[0].strip() for l in open(os.path.join(ROOT, script)).read().split("\n") if l.strip() and not l.strip().startswith("#")]   # (commands without their trailing comments)
        guard = [i for i, l in enumerate(lines) if "MOR_NO_AUTOBUILD=1" in l]
        prof = [i for i, l in enumerate(lines) if l.startswith("rocprofv3 ") or " rocprofv3 " in l]
        assert guard and prof and guard[0] < prof[0], script

