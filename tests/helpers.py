"""Shared helpers for the test-suite: load F1 fixtures and rebuild their seeded inputs."""
import ast
import glob
import os

import numpy as np

from oracle import forward_np as onp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def f1_names():
    return sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, "f1_*.npz")))


def load_f1(name):
    """-> dict(cfg, w, inputs, states, logits, probs, inter, n). Regenerates seeded tensors and verifies
    the checksums stored with the fixture (guards against generator drift)."""
    d = np.load(os.path.join(GOLDEN, "f1_%s.npz" % name))
    cfg = onp.OracleConfig(**ast.literal_eval(str(d["cfg"])))
    n = int(d["n"])
    w = onp.make_weights(cfg, int(d["wseed"]), float(d["wscale"]))
    wsum = float(sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values()))
    assert abs(wsum - float(d["wsum"])) <= 1e-9 * abs(wsum), "weight generator drifted"
    if "extreme" in name:
        inputs = onp.make_extreme_inputs(cfg, n, int(d["iseed"]))
    else:
        inputs = onp.make_inputs(cfg, n, int(d["iseed"]), wide_alphabet=("wide" in name))
    if "isum" in d.files:
        isum = float(sum(np.abs(a.astype(np.float64)).sum() for a in inputs))
        assert abs(isum - float(d["isum"])) <= 1e-9 * abs(isum), "input generator drifted"
    if "sseed" in d.files:
        states = onp.make_init_states(cfg, n, int(d["sseed"]))
        ssum = float(sum(np.abs(a.astype(np.float64)).sum() for a in states.values()))
        assert abs(ssum - float(d["ssum"])) <= 1e-9 * abs(ssum), "state generator drifted"
    else:
        states = {k[6:]: d[k] for k in d.files if k.startswith("state_")}
    inter = {k[6:]: d[k] for k in d.files if k.startswith("inter_")}
    return dict(cfg=cfg, w=w, inputs=inputs, states=states, logits=d["logits"], probs=d["probs"], inter=inter,
                n=n, raw=d)


def noise_floor(name):
    """max |dprob| between the reference's fp32 output and the float64 evaluation of the same model, as stored with the
    fixture by make_golden.py (round 3 fixtures), else None"""
    d = np.load(os.path.join(GOLDEN, "f1_%s.npz" % name))
    return float(d["f64_dprob"]) if "f64_dprob" in d.files else None


def f1_tolerances(name):
    """(oracle-vs-reference, HIP-vs-reference) bounds on |dprob| for fixture `name`.  Default: 1e-6 for the CPU
    restatements and 2e-5 for the HIP path (regression guards far inside the 1e-4 contract).  The saturating-weight
    fixtures (x5, x8) amplify fp32 summation-order differences: the reference's own fp32 result is 8e-5 away from the
    float64 evaluation of the same model on both_x8_n96 (make_golden.py printout), so there only the contract itself
    (1e-4, BASELINE.json north_star) can be asserted, for the oracle and for the HIP path alike."""
    if "_x8" in name:
        return 1e-4, 1e-4
    if "_x6p5" in name or "_x7" in name:
        # round 3 ladder between x5 and x8: the fixture stores the reference's own fp32-vs-float64 distance (f64_dprob,
        # 1.5e-5 / 1.7e-5); another fp32 evaluation order may sit that far on the other side of the float64 value
        return 2.5 * noise_floor(name), min(1e-4, 4.0 * noise_floor(name))
    if "_x5" in name:
        return 1e-5, 1e-4
    if "_x4" in name or "_x3" in name:
        return 2e-6, 2e-5
    return 1e-6, 2e-5


F8_CKPT = os.path.join(GOLDEN, "f8_trained_h128.ckpt")
F8_ROWS = os.path.join(GOLDEN, "f8_trained_rows.tsv")
# the same for the reference's DEFAULT architecture (hid_rnn 256): an 18.8 MB checkpoint (make_golden_trained.py --hid_rnn
# 256), committed since round 4 so that a fresh clone runs the default-architecture trained-weights tests instead of
# skipping them; other sizes would be generated into tests/golden/local/ (git-ignored)
F8_LOCAL = os.path.join(GOLDEN, "local")


def f8_paths(hid=128):
    if hid == 128:
        return F8_CKPT, os.path.join(GOLDEN, "f8_trained_expected.npz")
    if hid == 256:
        return os.path.join(GOLDEN, "f8_trained_h256.ckpt"), os.path.join(GOLDEN, "f8_trained_h256_expected.npz")
    return os.path.join(F8_LOCAL, "f8_trained_h%d.ckpt" % hid), os.path.join(F8_LOCAL, "f8_trained_h%d_expected.npz" % hid)


def have_f8(hid=128):
    return all(os.path.exists(p) for p in f8_paths(hid))


def load_f8(hid=128):
    """F8 (tests/golden/make_golden_trained.py): a checkpoint the reference's own `train` wrote, 400 labelled rows, and the
    reference model's outputs on them.  -> dict(cfg, w, inputs, states, logits, probs, logits0, probs0, labels, noise, raw);
    inputs come from THIS build's row parser (the reference parsed the same text with dataloader.parse_a_line2)."""
    import torch
    from deepsignal_plant_amd import textio
    ckpt, expected = f8_paths(hid)
    d = np.load(expected)
    cfg = onp.OracleConfig(**ast.literal_eval(str(d["cfg"])))
    sd = torch.load(ckpt, map_location="cpu")
    w = {k: np.ascontiguousarray(v.numpy().astype(np.float32)) for k, v in sd.items()}
    rows = textio.parse_rows(open(F8_ROWS, "rb").read(), cfg.seq_len, cfg.signal_len)
    inputs = [rows.kmer.astype(np.float32), rows.means, rows.stds, rows.lens.astype(np.float32), rows.signals]
    n = int(d["n"])
    assert rows.n == n
    states = onp.make_init_states(cfg, n, int(d["sseed"]))
    return dict(cfg=cfg, w=w, inputs=inputs, states=states, logits=d["logits"], probs=d["probs"],
                logits0=d["logits_zero_states"], probs0=d["probs_zero_states"], labels=d["labels"],
                row_labels=np.asarray(rows.labels), noise=float(d["f64_dprob"]), n=n, raw=d, ckpt=ckpt)


def rows_to_tsv(path, kmer, means, stds, lens, signals, labels=None):
    """feature-TSV rows (extract_features.py:381-395's grammar) whose parsed values are EXACTLY these float32 arrays: every
    number is printed with 9 significant digits, which round-trips a float32 through the parser's decimal -> double ->
    float32 conversion.  kmer: codes (any numeric dtype), lens: integers."""
    from deepsignal_plant_amd.utils.process_utils import code2base_dna
    n = len(kmer)
    f = lambda x: "%.9g" % float(np.float32(x))
    with open(path, "w") as wf:
        for i in range(n):
            cols = ["chr%d" % (i % 3 + 1), str(100 + i), "+", str(100 + i), "read_%d" % (i // 7), "t",
                    "".join(code2base_dna[int(c)] for c in kmer[i]),
                    ",".join(f(x) for x in means[i]), ",".join(f(x) for x in stds[i]),
                    ",".join(str(int(x)) for x in lens[i]),
                    ";".join(",".join(f(x) for x in row) for row in signals[i]),
                    str(int(labels[i]) if labels is not None else i % 2)]
            wf.write("\t".join(cols) + "\n")


def run_cli_jobs(tmp_path, jobs, world=1, timeout=900, env=None, tag="jobs"):
    """tests/cli_jobs.py: every job ({"argv": [...], "env": {...}}) in ONE process -- or one torchrun launch of `world` ranks --
    so that interpreter start, `import torch`, HIP initialisation (and the rendezvous) are paid once.  -> the per-job results
    [{"rc", "stdout", "stderr", "seconds", "error"}], plus the launcher's CompletedProcess as results.proc"""
    import json
    import socket
    import subprocess
    import sys
    path = os.path.join(str(tmp_path), "%s.json" % tag)
    with open(path, "w") as f:
        json.dump(jobs, f)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    e.update(env or {})
    if world > 1:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port), "-m", "tests.cli_jobs", path]
    else:
        cmd = [sys.executable, "-m", "tests.cli_jobs", path]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)

    class Results(list):
        pass
    out = Results(json.load(open(path + ".out")) if os.path.exists(path + ".out") else [])
    out.proc = proc
    return out


def cached_build(cmd, name, timeout=900):
    """Run the compiler command `cmd` + ["-o", <path>] once per state of the sources: the binary lives in
    tests/native/_build/<hash of the command and of every native source file>/<name> (git-ignored), so a second run of the suite
    on unchanged sources does not compile the sanitizer builds again.  Returns the binary's path."""
    import hashlib
    import subprocess
    h = hashlib.sha256(("\0".join(cmd) + "\0" + name).encode())
    for d in (os.path.join(ROOT, "deepsignal_plant_amd", "csrc"), os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native"),
              os.path.join(ROOT, "tests", "native", "emu"), os.path.join(ROOT, "tests", "native", "emu", "hip"), os.path.join(ROOT, "oracle")):
        for f in sorted(os.listdir(d)):
            path = os.path.join(d, f)
            if os.path.isfile(path) and f.endswith((".cpp", ".h", ".hip", ".c", ".hpp")):
                h.update(f.encode())
                with open(path, "rb") as fh:
                    h.update(fh.read())
    d = os.path.join(ROOT, "tests", "native", "_build", h.hexdigest()[:16])
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, name)
    with _BUILD_GUARD:
        lock = _BUILD_LOCKS.setdefault(out, threading.Lock())
    with lock:      # (background jobs of one session may ask for the same binary at once: one of them compiles it)
        if not os.path.exists(out):
            tmp = out + ".tmp%d.%d" % (os.getpid(), threading.get_ident())
            r = subprocess.run(list(cmd) + ["-o", tmp], capture_output=True, text=True, timeout=timeout)
            assert r.returncode == 0, r.stderr[-4000:]
            os.replace(tmp, out)
    return out


import threading  # noqa: E402

_BUILD_GUARD = threading.Lock()
_BUILD_LOCKS = {}
