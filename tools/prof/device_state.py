"""Clock / power / partition snapshot of GPU 0 through the SMI command-line tools, for bench.py's
`device_state` block: which kind of box produced a line (the pool hands out two kinds that differ by
1.6x on latency-bound kernels) and what the part was doing under the bench's loads.
Measurement only; returns {} where no SMI tool answers."""
import json
import subprocess


def _run(cmd, timeout=20):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        return r.stdout if r.returncode == 0 else ""
    except Exception:
        return ""


def _json_tail(text):
    at = text.find("{")
    if at < 0:
        at = text.find("[")
    if at < 0:
        return None
    try:
        return json.loads(text[at:])
    except Exception:
        return None


def _first_gpu(doc):
    if isinstance(doc, dict) and "gpu_data" in doc:
        doc = doc["gpu_data"]
    if isinstance(doc, list) and doc:
        return doc[0]
    return doc if isinstance(doc, dict) else {}


def _value(x):
    return x.get("value") if isinstance(x, dict) else x


def static_info():
    """identity of the part: serial number (tells boxes apart), partition modes, VBIOS"""
    out = {}
    g = _first_gpu(_json_tail(_run(["amd-smi", "static", "--asic", "--vbios", "--json"])) or {})
    asic = g.get("asic", {})
    for k in ("market_name", "asic_serial", "oam_id", "num_compute_units"):
        if k in asic:
            out[k] = asic[k]
    vb = g.get("vbios") or g.get("ifwi") or {}
    for k in ("version", "part_number", "build_date"):
        if isinstance(vb, dict) and k in vb:
            out["vbios_" + k] = vb[k]
    text = _run(["rocm-smi", "--showcomputepartition", "--showmemorypartition", "--showperflevel"])
    for ln in text.splitlines():
        for key, name in (("Compute Partition:", "compute_partition"),
                          ("Memory Partition:", "memory_partition"),
                          ("Performance Level:", "performance_level")):
            if key in ln and ln.startswith("GPU[0]"):
                out[name] = ln.split(key)[1].strip()
    return out


def sample():
    """shader clock of every XCD, memory / fabric clock and socket power right now"""
    out = {}
    g = _first_gpu(_json_tail(_run(["amd-smi", "metric", "--clock", "--power", "--json"])) or {})
    clocks = g.get("clock", {})
    gfx = [_value(v.get("clk")) for k, v in sorted(clocks.items()) if k.startswith("gfx_") and
           isinstance(v, dict)]
    gfx = [c for c in gfx if isinstance(c, (int, float))]
    if gfx:
        out["gfx_clk_mhz"] = gfx
    for name in ("mem_0", "fclk_0", "soc_0"):
        v = clocks.get(name)
        if isinstance(v, dict) and isinstance(_value(v.get("clk")), (int, float)):
            out[name.split("_")[0] + "_clk_mhz"] = _value(v.get("clk"))
    power = g.get("power", {})
    if isinstance(_value(power.get("socket_power")), (int, float)):
        out["socket_power_w"] = _value(power.get("socket_power"))
    if isinstance(power.get("throttle_status"), str):
        out["throttle_status"] = power["throttle_status"]
    if not out:  # fall back to rocm-smi's text
        for ln in _run(["rocm-smi", "--showclocks", "--showpower"]).splitlines():
            if not ln.startswith("GPU[0]"):
                continue
            if "sclk clock level" in ln and "(" in ln:
                out["gfx_clk_mhz"] = [float(ln.split("(")[1].split("Mhz")[0])]
            if "Package Power" in ln:
                out["socket_power_w"] = float(ln.split(":")[-1])
    return out


if __name__ == "__main__":
    print(json.dumps({"static": static_info(), "now": sample()}))
