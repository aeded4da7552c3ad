"""Seeded slices of the randomised sweeps in tools/ (fuzz_parity.py: host API against the oracle; fuzz_batched.py: the batched
device-resident call against the host API) - sizes, scenes, parameters and modes nobody wrote a dedicated case for."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_random_configurations_of_the_host_api_equal_the_oracle(capsys):
    rc = _tool("fuzz_parity").main(["--n", "60", "--seed", "11", "--budget-s", "120"])
    out = capsys.readouterr().out
    assert rc == 0 and "MISMATCH" not in out and "refused" not in out, out[-2000:]
    assert " 0 with a mismatch" in out


def test_random_configurations_of_the_batched_call_equal_the_host_api(capsys):
    import torch
    try:
        rc = _tool("fuzz_batched").main(["--n", "120", "--seed", "12", "--budget-s", "120"])
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream(torch.device("cuda", 0)))
    out = capsys.readouterr().out
    assert rc == 0 and "MISMATCH" not in out and "refused" not in out, out[-2000:]
    assert " 0 with a mismatch" in out


def test_random_configurations_of_the_geometry_calls_equal_the_oracle(capsys):
    rc = _tool("fuzz_geometry").main(["--n", "80", "--seed", "13", "--budget-s", "120"])
    out = capsys.readouterr().out
    assert rc == 0 and "MISMATCH" not in out, out[-2000:]
    assert " 0 with a mismatch" in out


def test_random_sequences_streamed_equal_the_per_frame_loop(capsys):
    """tools/fuzz_frames.py: FrameStream against the single-frame calls with pair_index (token, upload and stale-token routes), the fused
    initialisation step against matcher + two-view call - random sizes, detectors, chunk sizes, sequence lengths"""
    rc = _tool("fuzz_frames").main(["--n", "40", "--seed", "14", "--budget-s", "120"])
    out = capsys.readouterr().out
    assert rc == 0 and "MISMATCH" not in out and "refused" not in out, out[-2000:]
    assert " 0 with a mismatch" in out
