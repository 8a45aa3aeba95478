#!/usr/bin/env python3
"""Static instruction counts of the accumulation loop of every k_accumulate<curve> instantiation,
from the gfx950 ISA hipcc emits (no GPU needed).

    python tools/prof/isa_count.py [--out profiles/isa_counts.json] [--keep-asm DIR] [extra hipcc flags]

Every msm_<curve>.hip translation unit is compiled with `-S --offload-device-only` (the flags of
blitzar_amd/build.py); inside each k_accumulate function the main loop is the first backward-branch
region (in code order) that contains multiplies (the per-entry loop: the empty-bucket loop and the
flush block sit inside it, and are counted with it -- statically, i.e. as if taken on every iteration; the PMC count SQ_INSTS_VALU per
launch / (additions / 64) is the dynamic figure).  One loop iteration = one bucket addition per lane.
Reported per kernel: instructions of the loop by class, VGPRs, scratch bytes.
bench.py reads `mads_per_addition` from the JSON for the integer-ALU side of the roofline.
"""
import collections
import concurrent.futures
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blitzar_amd import build as bz_build  # noqa: E402  (FLAGS, HIPCC)

TUS = {"curve25519": "msm/msm_curve25519_accumulate.hip",
       "curve25519_niels": "msm/msm_curve25519_niels_accumulate.hip",
       "bls12_381": "msm/msm_bls12_381_accumulate.hip", "bn254": "msm/msm_bn254_accumulate.hip",
       "grumpkin": "msm/msm_grumpkin_accumulate.hip"}


def classify(op):
    if op.startswith("v_mad_u64_u32"):
        return "v_mad_u64_u32"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32")):
        return "v_mul_lo/hi_u32"
    if op.startswith(("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_lshl_add_u64",
                      "v_add_co", "v_addc_co", "v_add_u64")):
        return "valu_64bit_shift_add"
    if op.startswith(("v_and_b32", "v_bfe_u32", "v_and_or", "v_bfi")):
        return "valu_mask"
    if op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr", "v_cndmask", "v_readlane",
                      "v_writelane", "v_readfirstlane")):
        return "valu_move_select"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


# Issue cost per wave-instruction and SIMD by class, in shader cycles at 8 waves per SIMD
# (profiles/round2_valu_rates.txt, tools/ubench/valu_rates.hip): the 64-bit multiply-add, the other
# instructions of the multiplier / 64-bit shifter class, plain 32-bit VALU.
ISSUE_CYCLES = {"mad64": 4.47, "wide": 4.14, "plain": 2.37}
WIDE_OPS = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_add_co",
            "v_addc_co", "v_add_u64", "v_alignbit_b32", "v_mul_lo_u32", "v_mul_hi_u32",
            "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_mad_i32_i24", "v_mul_hi_i32")


def issue_class(op):
    if op.startswith("v_mad_u64_u32"):
        return "mad64"
    if op.startswith(WIDE_OPS):
        return "wide"
    if op.startswith("v_"):
        return "plain"
    return None


def analyse(asm_path):
    text = open(asm_path).read().split("\n")
    out = {}
    i = 0
    meta = {}
    # register / scratch use from the .amdhsa metadata comments hipcc leaves per kernel
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", "\n".join(text), re.S):
        body = m.group(2)
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        acc = re.search(r"\.amdhsa_accum_offset (\d+)", body)
        sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        meta[m.group(1)] = {"next_free_vgpr": int(vg.group(1)) if vg else None,
                            "accum_offset": int(acc.group(1)) if acc else None,
                            "scratch_bytes": int(sc.group(1)) if sc else None}
    while i < len(text):
        m = re.match(r"^(_ZN2bz12k_accumulateI\S+):", text[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        while j < len(text) and not text[j].startswith(".Lfunc_end"):
            j += 1
        body = text[i + 1:j]
        labels = {}
        for k, line in enumerate(body):
            lm = re.match(r"^(\.LBB\d+_\d+):", line)
            if lm:
                labels[lm.group(1)] = k
        # backward branches = loops; the per-entry loop is the first one in code order that holds
        # multiplies (the fold of whole-bucket runs behind it is a loop around one complete addition)
        best = None
        for k, line in enumerate(body):
            bm = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", line)
            if bm and bm.group(1) in labels and labels[bm.group(1)] < k:
                lo = labels[bm.group(1)]
                if not any("v_mad_u64_u32" in x for x in body[lo:k]):
                    continue
                if best is None or lo < best[0] or (lo == best[0] and k > best[1]):
                    best = (lo, k)
        counts = collections.Counter()
        whole = collections.Counter()
        issue = collections.Counter()
        for k, line in enumerate(body):
            om = re.match(r"^\s+([a-z_0-9]+)", line)
            if not om or line.lstrip().startswith((";", ".")):
                continue
            c = classify(om.group(1))
            whole[c] += 1
            if best and best[0] <= k <= best[1]:
                counts[c] += 1
                ic = issue_class(om.group(1))
                if ic:
                    issue[ic] += 1
        valu = sum(v for c, v in counts.items() if c.startswith(("v_", "valu")))
        demangled = subprocess.run(["c++filt", name], capture_output=True,
                                   text=True).stdout.strip()
        short = re.sub(r"^void bz::(k_accumulate<bz::\w+>).*", r"\1", demangled)
        out[short] = {"loop_instructions": dict(sorted(counts.items())), "loop_valu": valu,
                      "mads_per_addition": counts.get("v_mad_u64_u32", 0),
                      # issue bound of the loop's own instruction mix, cycles per wave-addition
                      "issue_classes": dict(sorted(issue.items())),
                      "issue_cycles_per_addition": round(
                          sum(ISSUE_CYCLES[c] * v for c, v in issue.items()), 1),
                      "mad_issue_cycles_per_addition": round(
                          ISSUE_CYCLES["mad64"] * issue.get("mad64", 0), 1),
                      "loop_total": sum(counts.values()),
                      "kernel_total": sum(whole.values()), **meta.get(name, {})}
        i = j
    return out


def compile_tu(args):
    tu, flags, outdir = args
    dst = os.path.join(outdir, os.path.basename(tu).replace(".hip", ".s"))
    # with the flags the library is built with for this translation unit (build.py, TU_FLAGS)
    subprocess.run([bz_build.HIPCC, *bz_build.FLAGS, *bz_build.TU_FLAGS.get(tu, []), *flags, "-S",
                    "--offload-device-only",
                    os.path.join(bz_build.CSRC, tu), "-o", dst], check=True)
    return dst


def main():
    argv = sys.argv[1:]
    out_path = None
    keep = None
    flags = []
    only = None
    k = 0
    while k < len(argv):
        if argv[k] == "--out":
            out_path = argv[k + 1]
            k += 2
        elif argv[k] == "--keep-asm":
            keep = argv[k + 1]
            k += 2
        elif argv[k] == "--only":
            only = argv[k + 1].split(",")
            k += 2
        else:
            flags.append(argv[k])
            k += 1
    outdir = keep or tempfile.mkdtemp(prefix="bz_isa_")
    os.makedirs(outdir, exist_ok=True)
    tus = {c: t for c, t in TUS.items() if only is None or c in only}
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(tus)) as ex:
        paths = list(ex.map(compile_tu, [(t, flags, outdir) for t in tus.values()]))
    result = {"source": "tools/prof/isa_count.py: hipcc -S --offload-device-only of "
                        "msm/msm_<curve>_accumulate.hip with the flags of blitzar_amd/build.py "
                        "(FLAGS + TU_FLAGS)" + (" + " + " ".join(flags) if flags else ""),
              "kernels": {}}
    for p in paths:
        result["kernels"].update(analyse(p))
    text = json.dumps(result, indent=1)
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(text + "\n")
    for name, r in result["kernels"].items():
        print(f"{name}: loop {r['loop_total']} instr, VALU {r['loop_valu']}, mads "
              f"{r['mads_per_addition']}, vgpr {r.get('next_free_vgpr')}, scratch {r.get('scratch_bytes')}")
        print("   ", r["loop_instructions"])


if __name__ == "__main__":
    main()
