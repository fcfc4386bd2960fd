"""Build-time rule for the direct (vector-ALU) kernels: no vector-memory load may have destination registers that overlap
its own address registers.  The compiler allows it; on a GPU shared by two processes such loads returned wrong data
(pointcloududa_amd/csrc/common.h, "VMEM address rule"; profiles/r02_two_process_determinism.txt).  These kernels have a
small LDS footprint and therefore share compute units with other processes' workgroups; they keep every in-flight
load's address alive with PCUDA_KEEP, and this test scans their device assembly for the pattern."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointcloududa_amd", "csrc")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles without a GPU)")
def test_direct_kernels_keep_load_addresses_alive():
    r = subprocess.run(["make", "-C", CSRC, "isa"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import vmem_overlap_scan as V
    rows = [r for r in V.scan(os.path.join(CSRC, "build", "isa")) if r[0] in ("conv_direct.s", "conv_direct_d1.s")]
    assert len(rows) >= 20, "expected the direct kernels' instantiations in the assembly"
    bad = [(f, k, n, ex) for f, k, n, ex in rows if n]
    assert not bad, "loads whose destination overlaps their address: %s" % bad[:4]


# loads of that form the compiler still emits in the LDS-free kernel families (one process per GPU never showed a wrong
# result from them; a GPU shared by two processes is an unsupported deployment: INTEGRATION.md section 4).  The counts
# are an allow-list: a compiler upgrade or a new kernel that ADDS such loads fails this test instead of going unnoticed.
# (pointwise.s: 11 until round 4; + 9 in the three pooled-source kernels, whose float2 loads of the pooled gradient keep the
# form although their addresses are held with PCUDA_KEEP)
ALLOWED = {"pointwise.s": 20, "losses.s": 15, "dense.s": 20, "sampler.s": 56, "conv1d_f32.s": 4, "optim.s": 0}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles without a GPU)")
def test_lds_free_kernel_families_do_not_grow_the_pattern():
    srcs = " ".join(k[:-2] + ".hip" for k in ALLOWED)
    r = subprocess.run(["make", "-C", CSRC, "isa", "ISA_SRCS=" + srcs], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import vmem_overlap_scan as V
    got = {}
    for f, _, n, _ in V.scan(os.path.join(CSRC, "build", "isa")):
        got[f] = got.get(f, 0) + n
    for f, lim in ALLOWED.items():
        assert f in got, "no assembly for " + f
        assert got[f] <= lim, "%s: %d loads whose destination overlaps their address (allow-list: %d)" % (f, got[f], lim)


def test_scanner_sees_the_pattern(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import vmem_overlap_scan as V
    (tmp_path / "k.s").write_text("_Z1kv:\n\tglobal_load_dwordx4 v[46:49], v[46:47], off\n\tglobal_load_dword v3, v[4:5], off\n"
                                  "\tglobal_load_dword v2, v2, s[12:13]\n\tglobal_load_dword v68, v[68:69], off offset:16\n\ts_endpgm\n")
    rows = V.scan(str(tmp_path))
    assert len(rows) == 1 and rows[0][2] == 3


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles without a GPU)")
def test_anti_phase_kernel_landing_zone_is_never_allocated(tmp_path):
    """csrc/conv_ap_impl.h keeps its input prefetch in flight across two barrier intervals in v[224:255], loaded and read back by
    inline assembly; the kernel is compiled with amdgpu_num_vgpr(224) so that the compiler never allocates those registers.
    A compiler that touched them (or spilled to AGPRs / scratch, which changes the register split) would corrupt the prefetch
    silently: scan every instantiation's assembly (scripts/ap_isa_check.py)."""
    r = subprocess.run(["make", "-C", CSRC, "isa", "ISA_SRCS=conv_ap.hip"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = os.path.join(CSRC, "build", "isa", "conv_ap.s")
    txt = open(s).read()
    assert txt.count("\n_Z14conv3ap_kernel") >= 8, "expected the kernel's eight instantiations in the assembly"
    assert "global_load_dwordx4 v[224:227]" in txt and "v_mov_b32 " in txt
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ap_isa_check.py"), s], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout[-2000:]
    # and the checker does see a violation
    bad = tmp_path / "bad.s"
    bad.write_text("_Z14conv3ap_kernelILi1ELb0ELb0ELb0ELb0EEv8ApParams:\n\tv_add_f32 v230, v1, v2\n\ts_endpgm\n")
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ap_isa_check.py"), str(bad)], capture_output=True, text=True)
    assert c.returncode == 1


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles without a GPU)")
def test_row_streaming_kernel_keeps_its_work_inside_the_mfma_sequence(tmp_path):
    """csrc/conv_rs.hip runs ONE wave per SIMD: only its own instruction order overlaps the conversion, the epilogue and the
    loads with the MFMAs (a scheduling fence pins one piece of that work behind every MFMA).  A compiler that clusters the MFMAs
    again, or spills, costs the kernel its overlap without changing a result: scan every shipped instantiation's assembly
    (scripts/rs_isa_check.py): no scratch in the variants the networks launch, 54 MFMAs per step copy, no long run of vector
    instructions without an MFMA."""
    r = subprocess.run(["make", "-C", CSRC, "isa", "ISA_SRCS=conv_rs.hip"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = os.path.join(CSRC, "build", "isa", "conv_rs.s")
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rs_isa_check.py"), s], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout[-2000:]
    # and the checker does see a clustered kernel
    bad = tmp_path / "bad.s"
    body = "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n" * (54 * 8) + "\tv_add_f32 v1, v2, v3\n" * 60
    bad.write_text("_ZN12_GLOBAL__N_114conv3rs_kernelILi1ELb0ELb1ELi0EEEvNS_8RsParamsE:\n" + body +
                   "\t.amdhsa_kernel x\n\t\t.amdhsa_private_segment_fixed_size 0\n\t.end_amdhsa_kernel\n")
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rs_isa_check.py"), str(bad)], capture_output=True, text=True)
    assert c.returncode == 1
