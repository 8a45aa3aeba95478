"""Round-5 additions: the in-run ALU probe, the power / clock trace summary, and a caller stream that
may only use a handful of compute units (the chunked sort's workers meet at a barrier inside one
launch: they must all be resident)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smi_trace_summary_slices_by_wall_clock():
    """tools/prof/smi_trace.py `summarize`: min / mean / max of socket power and shader clocks inside a
    window, the mean power from the energy accumulator (counts of 15.259 uJ), and the share of the
    firmware's accumulation ticks spent power-limited"""
    sys.path.insert(0, os.path.join(ROOT, "tools", "prof"))
    import smi_trace
    samples = []
    for k in range(100):  # 50 Hz for two seconds; 1000 W = 65.5e6 counts per second
        samples.append({"t": 1000.0 + 0.02 * k, "current_socket_power": 1000 + (k % 3),
                        "energy_accumulator": int(k * 0.02 * 1000.0 / 15.259e-6),
                        "gfxclks": [2200 + k % 5] * 8, "ppt_residency_acc": 40 * k,
                        "accumulation_counter": 100 * k})
    out = smi_trace.summarize(samples, 1000.5, 1001.5)
    assert out["samples"] == 51 and abs(out["hz"] - 50.0) < 0.5
    assert out["socket_power_w"] == {"min": 1000, "mean": pytest.approx(1001.0, abs=0.1), "max": 1002}
    assert out["sclk_mhz"]["min"] == 2200 and out["sclk_mhz"]["max"] == 2204
    assert out["energy_counter_mean_w"] == pytest.approx(1000.0, rel=1e-3)
    assert out["power_limited_share"] == pytest.approx(0.4, abs=1e-3)
    assert smi_trace.summarize(samples, 5000.0, 5001.0)["samples"] == 0


@pytest.mark.gpu
def test_mad_rate_probe(gpu_backend):
    """bzamd_probe_mad_rate: v_mad_u64_u32 wave-instructions per second over the device, effective
    shader clock, cycles per wave-instruction and SIMD -- inside what an MI355X can do"""
    lib = gpu_backend.load()
    out = (ctypes.c_double * 4)()
    assert lib.bzamd_probe_mad_rate(20.0, out) == 0
    rate, clock, cycles, load_ms = list(out)
    assert 1.5e9 < clock < 2.6e9, clock
    assert 3.8 < cycles < 5.5, cycles
    assert rate == pytest.approx(1024 * clock / cycles, rel=0.05)
    assert load_ms >= 20.0


MASKED_CHILD = r"""
import ctypes, json, sys
import numpy as np
import torch
sys.path.insert(0, {root!r})
from blitzar_amd import api
from tests import util
lib = api.load()
assert api.init(api.SXT_GPU_BACKEND, 0) == 0
hip = ctypes.CDLL("libamdhip64.so")
mask = (ctypes.c_uint32 * 8)(0xffff, 0, 0, 0, 0, 0, 0, 0)     # 16 of the 256 compute units
stream = ctypes.c_void_p()
assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), 8, mask) == 0
n = {n}
rng = np.random.default_rng(5)
one = rng.integers(0, 256, (1, 32), dtype=np.uint8)
skew = np.tile(one, (n, 1))
skew[::3] = rng.integers(0, 256, (len(skew[::3]), 32), dtype=np.uint8)   # two rows in three are equal
cols = [skew, rng.integers(0, 256, (n, 32), dtype=np.uint8)]
gens = util.generators_for(0, n)
g = torch.from_numpy(np.ascontiguousarray(util.api_generators(0, gens))).cuda()
d_cols = [torch.from_numpy(c).cuda() for c in cols]
desc = (api.sxt_sequence_descriptor * 2)()
for i, d in enumerate(d_cols):
    desc[i] = api.sxt_sequence_descriptor(32, n, d.data_ptr(), 0)
out = torch.zeros((2, 32), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
lib.bzamd_msm_device(0, ctypes.c_void_p(out.data_ptr()), 2, desc, ctypes.c_void_p(g.data_ptr()), stream)
assert hip.hipStreamSynchronize(stream) == 0
np.save({cols_file!r}, np.stack(cols))
print("RESULT" + json.dumps(out.cpu().numpy().tolist()))
"""


@pytest.mark.gpu
def test_caller_stream_with_a_cu_mask(oracle, tmp_path):
    """A caller's stream restricted to 16 compute units, skewed data (oversized bucket groups: the
    chunked path of pass 2, whose workers wait for each other inside one launch).  The engine sizes the
    workers from the stream's CU mask (two per CU); with the fixed 128 of round 4 the resident ones
    would spin for the others forever.  A child process under a timeout: a hang fails, it does not
    hang the suite."""
    n = 40000
    cols_file = str(tmp_path / "cols.npy")
    code = MASKED_CHILD.format(root=ROOT, n=n, cols_file=cols_file)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.array(json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("RESULT"))[6:]),
                   dtype=np.uint8)
    cols = np.load(cols_file)
    want = oracle.commit(0, [(cols[0], False), (cols[1], False)], util.generators_for(0, n))
    assert np.array_equal(got, want)


def test_bench_box_record_from_a_trace_summary():
    """bench.py `box_record`: `roofline.box` and the flat scalars the driver's record keeps, from what
    device_state + the SMI trace hand it; and nothing breaks when the trace has nothing"""
    import importlib.util
    import types
    torch_stub = None
    try:
        import torch  # noqa: F401  (bench.py imports it at module level)
    except Exception:  # pragma: no cover
        torch_stub = types.ModuleType("torch")
        sys.modules["torch"] = torch_stub
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class Lib:
        @staticmethod
        def bzamd_slow_instruction_fetch():
            return 1

    seq = {"samples": 120, "hz": 50.0, "socket_power_w": {"min": 1370, "mean": 1380.5, "max": 1390},
           "sclk_mhz": {"min": 2300, "mean": 2335.0, "max": 2360}, "energy_counter_mean_w": 1381.0,
           "power_limited_share": 0.7, "uclk_mhz": 2000.0, "temperature_mem_c": 55}
    state = {"trace": {"static": {"asic_serial": "0xABC", "power_cap_w": 1400.0},
                       "under_sequence_load": seq,
                       "under_lone_call_load": {"socket_power_w": {"min": 1000, "mean": 1040.0, "max": 1100}}},
             "sequence_leg": {"ms_per_step_last_2500": 0.985}}
    roof = {}
    bench.box_record(roof, state, {"sustained_ms": 0.99}, 1.42, Lib)
    assert roof["box_asic_serial"] == "0xABC" and roof["box_fetch_kind"] == "slow-fetch"
    assert roof["box_sclk_mhz_under_sequence"] == 2335.0 and roof["box_power_w_mean"] == 1380.5
    assert roof["box_power_w_energy_counter"] == 1381.0 and roof["box_lone_call_ms"] == 1.42
    assert roof["box_sustained_ms_per_step"] == 0.985 and roof["box_temperature_mem_c_under_sequence"] == 55
    assert roof["box"]["lone_leg"]["socket_power_w"]["mean"] == 1040.0
    assert all(not isinstance(v, (dict, list)) for k, v in roof.items() if k != "box")
    empty = {}
    bench.box_record(empty, {"trace": {}}, {}, 1.2, Lib)   # amdsmi unavailable: still a record
    assert empty["box_fetch_kind"] == "slow-fetch" and empty["box"]["socket_power_w"] is None
