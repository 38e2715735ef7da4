"""CPU: the C-ABI shared library loads and exports every symbol include/mc_hip.h declares."""
import ctypes
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = (ROOT / "include" / "mc_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(mc):
    syms = declared_symbols()
    assert len(syms) >= 18
    L = ctypes.CDLL(str(mc.LIB_PATH))
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mc_hip.h but not exported"
    assert sorted(mc.ABI_SYMBOLS) == syms


def test_abi_version(mc):
    assert mc.lib().mc_abi_version() == 4   # v4: the multi-device sweep (mc_march_sharded, mc_comm_*, mc_march_rank), mc_index_rebase as a state


def test_no_cpu_fallback(mc):
    """Without a GPU the library must refuse loudly, never compute on the host."""
    if mc.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(mc.McError) as e:
        mc.Context(0)
    assert e.value.code == mc.MC_ERR_HIP and "no CPU fallback" in str(e.value)


def test_jit_source_compiles_for_gfx950(mc):
    """hiprtc cross-compiles the specialised kernels without a GPU."""
    assert mc.jit_precompile("x^2+y^2+z^2-1") > 10000
    assert mc.jit_precompile("x^y+z^-3-(x/y)") > 10000


def test_product_does_not_link_oracle(mc):
    import subprocess
    out = subprocess.run(["ldd", str(mc.LIB_PATH)], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", str(mc.LIB_PATH)], capture_output=True, text=True).stdout
    assert "orc_" not in syms


def test_only_the_c_abi_is_exported(mc):
    """The library is built with -fvisibility=hidden: of the symbols it DEFINES, `nm -D` shows the entry points
    include/mc_hip.h declares and nothing else -- no C++ internals (mc::tokenize, mc::compile ...), no helpers."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", str(mc.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    defined = set()
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 3 and parts[1] in "TtWwVvBbDdRr":
            defined.add(parts[2])
    # what the toolchain itself adds to every shared object / HIP fat binary registration
    toolchain = {"_init", "_fini", "__bss_start", "_edata", "_end", "__hip_fatbin", "__hip_gpubin_handle"}
    ours = {s for s in defined if s not in toolchain and not s.startswith("__hip_") and not s.startswith("_ZTS") and not s.startswith("_ZTI")}
    assert ours == set(declared_symbols()), sorted(ours ^ set(declared_symbols()))


def test_header_is_plain_c(tmp_path):
    """include/mc_hip.h must be usable from C (the boundary is a C ABI): compile a C99 translation unit that uses every
    struct and calls every entry point's prototype, syntax only."""
    import subprocess
    from conftest import ROOT
    src = tmp_path / "use.c"
    src.write_text('#include "mc_hip.h"\n'
                   'int use(mc_context *c) {\n'
                   '    mc_params p = {"x+y", 0.25f, 0.0f, {1.0f, 1.0f, 1.0f}, MC_FLAG_NORMALS, 0, -1};\n'
                   '    mc_result r;\n'
                   '    mc_set_constraint(c, 0, "x", ">", -0.5f); mc_use_constraint(c, 0, 1); mc_set_seed(c, 0, 0, 0); mc_seed_mode(c, 0);\n'
                   '    return mc_march(c, &p, &r) + (int)r.n_tris + mc_abi_version();\n'
                   '}\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", f"-I{ROOT / 'include'}", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
