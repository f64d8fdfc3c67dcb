"""bench.py's command-line contract (no GPU needed): the driver calls `python bench.py --gpus N --steps K --warmup W`
and parses ONE JSON line; the keys that line must carry are spelled in the source."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_help_lists_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_json_line_has_the_contract_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert re.search(r'"%s"\s*[:\]]' % key, src), key
    for key in ("bound", "achieved", "peak", "frac", "traffic"):
        assert '"%s"' % key in src, key
    for key in ("cores", "kind", "sample"):
        assert '"%s"' % key in src, key
    assert '"workload"' in src and '"vs_baseline": None' in src


def test_without_a_gpu_it_refuses_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)
