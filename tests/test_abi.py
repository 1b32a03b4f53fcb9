"""The C-ABI library: loads, exports every symbol include/direct_ddp.h declares, ctypes mirrors have
the C sizes, and without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from direct_amd import abi, problems, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "direct_ddp.h")


HEADER_CLUSTER = os.path.join(ROOT, "include", "direct_cluster.h")


def declared_functions(header=HEADER):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(direct_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    lib = solver.lib()
    names = declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "libdirect_ddp.so does not export %s" % n
    assert set(names) == set(solver.EXPORTS)
    assert lib.direct_ddp_abi_version() == 1
    from direct_amd import cluster
    cnames = declared_functions(HEADER_CLUSTER)
    for n in cnames:
        assert hasattr(lib, n), "libdirect_ddp.so does not export %s" % n
    assert set(cnames) == set(cluster.EXPORTS)
    from direct_amd import quad
    qnames = declared_functions(os.path.join(ROOT, "include", "direct_quad.h"))
    for n in qnames:
        assert hasattr(lib, n), "libdirect_ddp.so does not export %s" % n
    assert set(qnames) == set(quad.EXPORTS)


def test_struct_sizes_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(direct_ddp_params_t),sizeof(direct_ddp_batch_in_t),sizeof(direct_ddp_batch_out_t),'
                   'sizeof(direct_ddp_config_t),sizeof(direct_sample_in_t),sizeof(direct_sample_out_t));return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(abi.Params), C.sizeof(abi.BatchIn), C.sizeof(abi.BatchOut), C.sizeof(abi.Config),
                     C.sizeof(abi.SampleIn), C.sizeof(abi.SampleOut)]
    from direct_amd import cluster
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu\\n",sizeof(direct_cluster_config_t),'
                   'sizeof(direct_rccl_id_t));return 0;}\n' % HEADER_CLUSTER)
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    assert [int(x) for x in subprocess.check_output([str(exe)]).split()] == [C.sizeof(cluster.Config), 128]
    from direct_amd import quad
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu\\n",sizeof(direct_quad_params_t));return 0;}\n'
                   % os.path.join(ROOT, "include", "direct_quad.h"))
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    assert int(subprocess.check_output([str(exe)])) == C.sizeof(quad.Params)


def _has_gpu():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=30).stdout
        return "gfx950" in out
    except Exception:
        return False


def test_no_cpu_fallback_without_device(built):
    if _has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(solver.DirectError) as e:
        solver.DdpSolver(4, 5, 6, np.float32)
    assert e.value.status == abi.DIRECT_ERR_NO_DEVICE
    from direct_amd import cluster
    with pytest.raises(solver.DirectError) as e:
        cluster.ClusterGenerator((8, 8, 8))
    assert e.value.status == abi.DIRECT_ERR_NO_DEVICE


def test_product_package_does_not_import_the_oracle():
    """direct_amd/ must never touch oracle/ or the emulator (they are test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "direct_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("oracle/ or", "") or f == "abi.py", (dirpath, f)
                assert "DIRECT_EMULATE" not in txt or f == "ddp_wave.h", (dirpath, f)


def test_tools_do_not_use_the_oracle():
    """tools/ are measurement helpers of the product path; scripts that need the oracle live under tests/soak/."""
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh", ".hip")):
            assert "oracle" not in open(os.path.join(ROOT, "tools", f)).read(), f


def test_create_rejects_bad_configs(built):
    lib = solver.lib()
    h = C.c_void_p()
    for cfg, want in ((abi.Config(5, 0, 4, 4, 6, 0), abi.DIRECT_ERR_INVALID),
                      (abi.Config(abi.F32, 0, 0, 4, 6, 0), abi.DIRECT_ERR_INVALID),
                      (abi.Config(abi.F32, 0, 4, 4, abi.P_LIMIT + 1, 0), abi.DIRECT_ERR_UNSUPPORTED)):
        assert lib.direct_ddp_create(C.addressof(cfg), C.addressof(h)) == want
        assert len(lib.direct_ddp_last_error()) > 0


def test_cluster_create_rejects_maps_whose_summed_area_table_passes_4_gib(built):
    """box_obstacles addresses the (X+1)(Y+1)(Z+1)-entry summed-area table with unsigned 32-bit BYTE offsets: a map that
    large must be refused at create time, not read wrong entries.  Sizes are validated before the device."""
    from direct_amd import cluster
    lib = solver.lib()
    h = C.c_void_p()
    for dims, want in (((1024, 1024, 1023), abi.DIRECT_ERR_UNSUPPORTED), ((1024, 1024, 1030), abi.DIRECT_ERR_UNSUPPORTED),
                       ((0, 10, 10), abi.DIRECT_ERR_INVALID)):
        cfg = cluster.Config(0, dims[0], dims[1], dims[2], 4, 4096, 4096, 0)
        assert lib.direct_cluster_create(C.addressof(cfg), C.addressof(h)) == want, dims


def test_time_allocation_through_the_abi(built):
    """initTimeAllocation (teach_repeat_planner.cpp:583-639) is host code in the library."""
    batch = problems.make_batch("free", 5, 9, seed=4)
    T = solver.time_allocation(batch.n_seg, batch.x0[:, :3], batch.xd[:, :3], batch.seeds)
    assert np.allclose(T, batch.T0, rtol=1e-14)


def test_algorithmic_words_formula():
    """257 + 5 nc + 8 P words per knot-iteration (feasible), P = 6 -> 760."""
    b = problems.make_batch("free", 2, 10, seed=1)
    assert problems.algorithmic_words(b.n_planes, b.n_seg) == 2 * 10 * 760
    assert problems.algorithmic_words(b.n_planes, b.n_seg, infeasible=True) == 2 * 10 * (257 + 910 + 48)


def test_row_dealing_reciprocal_table_is_exact():
    """kInvP of csrc/ddp_tables.h replaces r / P in the row -> (control point, plane) dealing."""
    import re
    txt = open(os.path.join(ROOT, "direct_amd", "csrc", "ddp_tables.h")).read()
    body = re.search(r"kInvP\[141\]\s*=\s*\{([^}]*)\}", txt).group(1)
    tab = [int(t) for t in body.replace("\n", " ").split(",")]
    assert len(tab) == 141 and "kInvPShift = 20" in txt
    for P in range(1, 141):
        for r in range(900):
            assert (r * tab[P]) >> 20 == r // P and r * tab[P] < 2 ** 32
