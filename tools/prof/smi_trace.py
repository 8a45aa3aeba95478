"""Power / clock trace of GPU 0 through the amdsmi library, in a process of its own.

    python tools/prof/smi_trace.py --hz 50 --out trace.jsonl     # until stdin closes or SIGTERM

bench.py starts it before its load legs and slices the samples by wall-clock windows afterwards
(`summarize`): socket power {min, mean, max}, shader clock per XCD, and -- independent of the sampling
rate -- the mean power from the part's own energy accumulator (delta energy / delta time).  One
library call per sample (amdsmi_get_gpu_metrics_info, ~1 ms), against the ~1 s a sample through the
amd-smi command line takes.  Measurement only; every failure is reported in the output, never raised.
"""
import argparse
import json
import os
import signal
import sys
import threading
import time

ENERGY_UJ_PER_COUNT = 15.259  # amdsmi energy_accumulator resolution on MI300-class parts (uJ)


def _num(x):
    return x if isinstance(x, (int, float)) and not isinstance(x, bool) else None


def open_gpu():
    import amdsmi
    amdsmi.amdsmi_init()
    handles = amdsmi.amdsmi_get_processor_handles()
    if not handles:
        raise RuntimeError("amdsmi: no processor handles")
    return amdsmi, handles[0]


def static_info(amdsmi, h):
    out = {}
    try:
        asic = amdsmi.amdsmi_get_gpu_asic_info(h)
        for k in ("market_name", "asic_serial", "oam_id", "num_compute_units"):
            if k in asic:
                out[k] = asic[k]
    except Exception as exc:
        out["asic_error"] = repr(exc)[:120]
    try:
        cap = amdsmi.amdsmi_get_power_cap_info(h)
        for k in ("power_cap", "max_power_cap", "default_power_cap"):
            if _num(cap.get(k)) is not None:
                out[k + "_w"] = cap[k] / 1e6 if cap[k] > 100000 else cap[k]
    except Exception as exc:
        out["power_cap_error"] = repr(exc)[:120]
    return out


def sample(amdsmi, h):
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    s = {"t": time.time()}
    for k in ("current_socket_power", "average_socket_power", "energy_accumulator",
              "current_gfxclk", "average_gfxclk_frequency", "ppt_residency_acc",
              "socket_thm_residency_acc", "accumulation_counter", "firmware_timestamp",
              "temperature_hotspot", "temperature_mem"):
        v = _num(m.get(k))
        if v is not None:
            s[k] = v
    clks = m.get("current_gfxclks")
    if isinstance(clks, (list, tuple)):
        s["gfxclks"] = [c for c in clks if _num(c) is not None]
    # memory and fabric-side clocks: the two kinds of boxes in the pool differ on everything that goes
    # through the L2 / fabric (instruction fetch beyond the I-cache, the memory-bound front kernels)
    for k in ("current_uclk", "current_socclk"):
        v = _num(m.get(k))
        if v is not None:
            s[k] = v
    socs = m.get("current_socclks")
    if isinstance(socs, (list, tuple)):
        socs = [c for c in socs if _num(c) is not None]
        if socs:
            s["socclks"] = socs
    return s


def run(args):
    stop = threading.Event()
    signal.signal(signal.SIGTERM, lambda *_: stop.set())
    signal.signal(signal.SIGINT, lambda *_: stop.set())

    def watch_stdin():  # the parent closes our stdin (or dies): stop
        try:
            sys.stdin.read()
        except Exception:
            pass
        stop.set()

    threading.Thread(target=watch_stdin, daemon=True).start()
    with open(args.out, "w") as fh:
        try:
            amdsmi, h = open_gpu()
        except Exception as exc:
            fh.write(json.dumps({"error": "amdsmi unavailable: " + repr(exc)[:200]}) + "\n")
            return 1
        fh.write(json.dumps({"static": static_info(amdsmi, h), "hz": args.hz}) + "\n")
        fh.flush()
        period = 1.0 / args.hz
        nxt = time.time()
        deadline = time.time() + args.max_seconds
        while not stop.is_set() and time.time() < deadline:
            try:
                fh.write(json.dumps(sample(amdsmi, h)) + "\n")
            except Exception as exc:
                fh.write(json.dumps({"t": time.time(), "error": repr(exc)[:120]}) + "\n")
            nxt += period
            delay = nxt - time.time()
            if delay > 0:
                stop.wait(delay)
            else:
                nxt = time.time()
        fh.flush()
    return 0


def load(path):
    static, samples, errors = {}, [], []
    if not os.path.exists(path):
        return static, samples, ["no trace file"]
    with open(path) as fh:
        for ln in fh:
            try:
                d = json.loads(ln)
            except Exception:
                continue
            if "static" in d:
                static = d["static"]
            elif "error" in d:
                errors.append(d["error"])
            elif "t" in d:
                samples.append(d)
    return static, samples, errors


def summarize(samples, t0, t1):
    """what the part did between wall-clock times t0 and t1"""
    win = [s for s in samples if t0 <= s["t"] <= t1]
    out = {"samples": len(win), "seconds": round(t1 - t0, 3)}
    if len(win) >= 2:
        out["hz"] = round((len(win) - 1) / max(win[-1]["t"] - win[0]["t"], 1e-9), 1)
    p = [s["current_socket_power"] for s in win if "current_socket_power" in s]
    if p:
        out["socket_power_w"] = {"min": min(p), "mean": round(sum(p) / len(p), 1), "max": max(p)}
    clk = [c for s in win for c in s.get("gfxclks", [])]
    if clk:
        out["sclk_mhz"] = {"min": min(clk), "mean": round(sum(clk) / len(clk), 1), "max": max(clk)}
    for key, name in (("current_uclk", "uclk_mhz"), ("current_socclk", "socclk_mhz")):
        v = [s[key] for s in win if key in s]
        if v:
            out[name] = round(sum(v) / len(v), 1)
    for key, name in (("temperature_hotspot", "temperature_hotspot_c"), ("temperature_mem", "temperature_mem_c")):
        v = [s[key] for s in win if key in s]
        if v:
            out[name] = max(v)
    soc = [c for s in win for c in s.get("socclks", [])]
    if soc:
        out["socclks_mhz_mean"] = round(sum(soc) / len(soc), 1)
    e = [(s["t"], s["energy_accumulator"]) for s in win if "energy_accumulator" in s]
    if len(e) >= 2 and e[-1][0] > e[0][0] and e[-1][1] > e[0][1]:
        out["energy_counter_mean_w"] = round(
            (e[-1][1] - e[0][1]) * ENERGY_UJ_PER_COUNT * 1e-6 / (e[-1][0] - e[0][0]), 1)
    r = [(s["t"], s["ppt_residency_acc"], s.get("accumulation_counter")) for s in win
         if "ppt_residency_acc" in s]
    if len(r) >= 2 and r[0][2] is not None and r[-1][2] is not None and r[-1][2] > r[0][2]:
        # share of the firmware's accumulation ticks spent limited by the package power tracker
        out["power_limited_share"] = round((r[-1][1] - r[0][1]) / (r[-1][2] - r[0][2]), 3)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--hz", type=float, default=50.0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--max-seconds", type=float, default=600.0)
    sys.exit(run(ap.parse_args()))
