"""GPU: buffer descriptors with real extents (round 6).  Named to run LAST in the suite: round 6 had no GPU, so
these tests have never run -- under the driver's `-x` a first-contact surprise in here must not cost the rest of the record.

  * the hardware does range-check: a load past num_records returns zeros, a store past it is dropped -- by VGPR offset AND by
    SGPR offset (where the kernels carry almost all of an address);
  * the product library gives the same bytes with the extents of the allocation (default) and the tight extents of the call
    (and the 2 GiB windows of rounds 1-5: tests/test_gpu_parity.py), over every kernel form (tools/extents_sweep.py: 11 model shapes, 45 batch sizes, four
    precisions, Philox / explicit / zero states; the suite runs the six shapes that take every kernel form);
  * the bounds-recording build (libdsp_amd_bounds.so: every descriptor access compared in software with the TIGHT extent of
    its operand) runs the same sweep without a record and with the same bytes;
  * negative control: with 4 KiB taken off every LSTM launch's input extent the bounds build returns DSP_EBOUNDS naming the
    operand, the source line, the workgroup and the offset."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu
GPU_MODULE_TIMEOUT = 3600      # (tests/gpu_isolation.py: three sweeps, one of them under the software-checked bounds build)
BOUNDS_LIB = os.path.join(ROOT, "deepsignal_plant_amd", "libdsp_amd_bounds.so")


def _sweep(env_extra, *cases, expect_rc=0):
    env = {k: v for k, v in os.environ.items() if k not in ("DSP_AMD_LIB", "DSP_RSRC_EXTENTS", "DSP_BOUNDS_TEST_SHRINK")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extents_sweep.py")] + list(cases), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == expect_rc, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return r


def _probe():
    from deepsignal_plant_amd import _native
    got = _native.range_probe(0)
    print("range probe (lanes in range that read their data, lanes past the extent by voffset that read 0, ... by soffset, floats untouched "
          "by 64 out-of-range stores):", got, "-- expected (16, 48, 64, 1024)")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "range_probe.json"), "w") as f:
        json.dump({"in_range_ok": got[0], "voffset_oob_zero": got[1], "soffset_oob_zero": got[2], "floats_untouched": got[3]}, f)
    return got


def test_accesses_inside_a_descriptors_extent_are_untouched_and_the_probe_does_not_fault():
    """what the product relies on unconditionally: a descriptor with a real num_records serves every access inside it"""
    got = _probe()
    assert got[0] == 16, got


def test_the_hardware_drops_accesses_past_the_extent_by_vgpr_offset_and_by_sgpr_offset():
    """what turns a wild offset into a parity failure instead of a dead process.  The kernels carry almost all of an address in
    the SGPR offset: if the hardware's check looked at the VGPR offset alone (got[2] < 64, or stores landing: got[3] < 1024),
    the extents would still be harmless -- they never clip a legitimate access -- but the protection would be partial, and
    DESIGN.md 3 has to say so."""
    got = _probe()
    assert got[0] == 16, got
    if got != (16, 48, 64, 1024):
        # a property of the hardware, observed here for the first time, that no result depends on: reported, not failed (a
        # failure would end the driver's `-x` run in front of the sweeps below, which do not depend on it)
        pytest.xfail("the range check of this device is partial: %r instead of (16, 48, 64, 1024) -- the extents stay harmless "
                     "(they never clip a legitimate access), the protection DESIGN.md 3 describes is weaker" % (got,))


# the shapes of tools/extents_sweep.py that between them take every kernel form (all eleven: `python tools/extents_sweep.py` by hand,
# tools/r6_final.sh) -- the suite's share of the GPU box's time stays at a few minutes
SHAPES = ("default", "cfg3_seq_only", "hid128", "hid320_many_pass", "hid100_ut4_padded", "s40_wide_window")


@pytest.fixture(scope="module")
def product_digest():
    return json.loads(_sweep({}, *SHAPES).stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("mode", ["tight"])   # ("wide" against the default: test_small_batch_kernels_do_not_change_a_bit too, five shapes x fifteen sizes)
def test_extents_do_not_change_a_bit(product_digest, mode):
    got = json.loads(_sweep({"DSP_RSRC_EXTENTS": mode}, *SHAPES).stdout.strip().splitlines()[-1])
    assert len(got) == len(product_digest) > 70
    diff = [k for k in product_digest if got.get(k) != product_digest[k]]
    assert not diff, diff[:10]


def test_the_bounds_build_runs_every_kernel_form_without_a_record_and_with_the_same_bytes(product_digest):
    assert os.path.exists(BOUNDS_LIB), "make -C deepsignal_plant_amd/csrc bounds (build() does)"
    got = json.loads(_sweep({"DSP_AMD_LIB": BOUNDS_LIB}, *SHAPES).stdout.strip().splitlines()[-1])
    diff = [k for k in product_digest if got.get(k) != product_digest[k]]
    assert len(got) == len(product_digest) and not diff, diff[:10]


def test_the_bounds_build_names_an_access_past_a_shortened_extent():
    r = _sweep({"DSP_AMD_LIB": BOUNDS_LIB, "DSP_BOUNDS_TEST_SHRINK": "4096"}, "default", expect_rc=3)
    assert "out of range" in r.stderr and "operand K4 input" in r.stderr and "dsp_kernels.hip:" in r.stderr, r.stderr[-2000:]
    print(r.stderr.strip().splitlines()[-1])


def test_x_ahead_does_not_change_a_bit_and_what_it_costs():
    """DSP_LSTM_XAHEAD=1 (opt-in; dsp_xahead_kernel + dsp_lstmc_kernel<.., XA>): the x part of the clustered dense layers summed
    ahead of the recurrence for calls of <= 256 sites.  Written and verified (bit-identical, under the SIMT interpreter:
    tests/test_kernel_emu.py) in a round without a GPU -- this is its first contact with hardware, hence the LAST test of the
    last module.  The bytes are held; the timings are printed and kept (gpurun_out/xahead_ab.json), not asserted."""
    env = {k: v for k, v in os.environ.items() if k not in ("DSP_AMD_LIB", "DSP_RSRC_EXTENTS", "DSP_LSTM_XAHEAD", "DSP_LSTM_XAHEAD_RING", "DSP_LSTM_XAHEAD_TILES", "DSP_LSTM_HANDOFF", "DSP_LSTM_CLUSTER")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "xahead_ab.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    got = json.loads(r.stdout.strip().splitlines()[-1])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "xahead_ab.json"), "w") as f:
        json.dump(got, f, sort_keys=True)
    for k, (off, on, on8) in sorted(got["ms"].items(), key=lambda kv: (kv[0].split("/")[0], int(kv[0].split("/")[1]))):
        print("%-32s %.4f -> %.4f ms per forward (%+.1f %%); rings 8 deep %.4f (%+.1f %%)" % (k, off, on, 100.0 * (on - off) / off, on8, 100.0 * (on8 - off) / off))
    assert got["cases"] >= 40 and got["identical"], got["differs"]
