"""CPU: the build-time gate for the gfx950 store-data hazard (tools/check_store_hazard.py; measured by
tools/micro/h_exchange.hip, profiles/r3/micro_h_exchange.txt): a wide VMEM store followed at once by a VALU write of its
data registers corrupts the store, and hipcc does not pad the SGPR-soffset form.  The scanner recognises the pattern (a
positive control on the sequences hipcc actually emitted) and the device assembly of every .hip translation unit of the
library is clean -- this is the test that catches the loss of the invariant the LSTM kernels' h exchange relies on."""
import glob
import os
import subprocess
import sys

from tests.helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_store_hazard as csh  # noqa: E402

CSRC = os.path.join(ROOT, "deepsignal_plant_amd", "csrc")


def _scan_text(tmp_path, text):
    p = tmp_path / "x.s"
    p.write_text("_Z1kv:\n" + text + "\ts_endpgm\n")
    found, stores = csh.scan(str(p))
    return [(f[6], f[8]) for f in found], stores   # (wait states before the writer, writer is VALU)


def test_scanner_recognises_the_measured_patterns(tmp_path):
    # what hipcc emitted for the micro-benchmark's check kernel, and what it emitted for the NP = 2 LSTM kernels in round 2
    bad, n = _scan_text(tmp_path, "\tbuffer_store_dwordx4 v[4:7], v17, s[8:11], s16 offen\n\tv_pk_add_f32 v[6:7], v[0:1], v[10:11] op_sel_hi:[0,1]\n")
    assert bad == [(0, True)] and n == 1
    bad, _ = _scan_text(tmp_path, "\tbuffer_store_dwordx4 v[0:3], v253, s[20:23], s0 offen\n\tv_mov_b32_e32 v0, 0\n")
    assert bad == [(0, True)]
    bad, _ = _scan_text(tmp_path, "\tbuffer_store_dwordx4 v[0:3], v253, s[20:23], s0 offen\n\ts_nop 0\n\tv_mov_b32_e32 v3, 0\n")
    assert bad == [(1, True)]                                   # one wait state: measured safe, below the margin
    for pad in ("\ts_nop 1\n", "\ts_nop 0\n\tv_add_f32_e32 v9, v8, v8\n", "\tv_mov_b32_e32 v8, 0\n\tv_mov_b32_e32 v9, 0\n"):
        bad, _ = _scan_text(tmp_path, "\tbuffer_store_dwordx4 v[0:3], v253, s[20:23], s0 offen\n" + pad + "\tv_mov_b32_e32 v1, 0\n")
        assert bad == []                                        # two wait states (the guard in dsp_kernels.hip: s_nop 1)
    bad, _ = _scan_text(tmp_path, "\tglobal_store_dwordx4 v[10:11], v[4:7], off\n\tv_mfma_f32_32x32x2_f32 v[0:15], v1, v2, v[0:15]\n")
    assert bad == [(0, True)]                                   # any VALU-class writer counts, MFMA included
    bad, _ = _scan_text(tmp_path, "\tscratch_store_dwordx4 off, v[0:3], off offset:16\n\tbuffer_load_dwordx4 v[0:3], v253, s[24:27], s15 offen\n")
    assert bad == [(0, False)]                                  # a memory return into the registers: reported, benign
    bad, _ = _scan_text(tmp_path, "\tbuffer_store_dwordx2 v[0:1], v253, s[20:23], s0 offen\n\tv_mov_b32_e32 v0, 0\n"
                                  "\tbuffer_store_dwordx4 v[0:3], v253, s[20:23], s0 offen\n\tv_mov_b32_e32 v9, 0\n\tv_cmp_gt_u32_e32 vcc, s29, v2\n")
    assert bad == []                                            # 64-bit stores and writers of other registers are fine
    assert csh.main([str(tmp_path / "x.s")]) == 0


def test_device_assembly_of_the_library_is_free_of_the_hazard():
    """make keeps the device assembly of every .hip translation unit under csrc/_obj (-save-temps=obj) and fails on the
    pattern; here the same scan runs on whatever is built (building first if need be)"""
    subprocess.check_call(["make", "-s", "-C", CSRC])
    files = sorted(glob.glob(os.path.join(CSRC, "_obj", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    names = {os.path.basename(f).split("-hip-")[0] for f in files}
    assert {"dsp_kernels", "dsp_extract", "dsp_freq_dev"} <= names, names
    total = 0
    for f in files:
        found, stores = csh.scan(f)
        total += stores
        assert [x for x in found if x[8]] == [], (f, [x[2:7] for x in found if x[8]][:3])
    assert total > 200      # the forward kernels alone hold > 200 wide stores: the scan really saw them
