"""CPU: bench.py's rank-launching logic (no GPU call is made by the launching process).  The 2-rank run itself is
tests/test_gpu_cli.py::test_bench_started_plainly_with_gpus_2_starts_two_ranks_itself (-m gpu)."""
import os
import sys

import pytest

from tests.helpers import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_n_without_a_launcher_starts_n_ranks_as_a_child(monkeypatch):
    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    argv = ["--gpus", "4", "--steps", "5", "--warmup", "1"]
    assert bench.main(argv) == 7                      # the child's exit code is this process's exit code
    (cmd, env), = calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_rank_or_a_single_gpu_run_does_not_launch(monkeypatch):
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert bench.self_launch(bench.parse_args(["--gpus", "1"]), []) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "2")
    assert bench.self_launch(bench.parse_args(["--gpus", "4"]), []) is None


def test_traffic_is_quoted_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    import json
    args = bench.parse_args([])
    h = bench.kernel_source_hash()
    entry = {"model_type": "both_bilstm", "layernum1": 3, "hid_rnn": 256, "batch": 65536, "precision": "fp32",
             "kernel_src_sha16": h, "hbm_bytes_per_launch": 1.25e10}
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps([entry]))
    assert bench.committed_traffic(args)["hbm_bytes_per_launch"] == 1.25e10
    assert bench.committed_traffic(args, stale=True) is None
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps([dict(entry, kernel_src_sha16="0" * 16)]))
    assert bench.committed_traffic(args) is None      # stale: the kernels changed since the PMC pass
    assert bench.committed_traffic(args, stale=True)["kernel_src_sha16"] == "0" * 16   # ... reported apart, under its own hash, never as `traffic`
    assert bench.committed_traffic(bench.parse_args(["--batch", "4096"])) is None
