#!/usr/bin/env python3
"""tools/isa_hist.py -- a per-phase instruction table of one kernel from its gfx950 assembly (round 6).

    python tools/isa_hist.py [--kernel REGEX] [--src starflate_amd/csrc/sf_kernels.hip] [--json out.json] [--blocks]

The source is compiled with `-S -gline-tables-only` (the line tables do not change the kernel's code; the tool checks the
instruction count against a build without them).  Every instruction of the kernel is attributed to a PHASE by its source
line: a phase begins at a marker comment

    // @phase <name> [trips=<float>] [note=...]

in the .hip source and lasts to the next marker; `trips` is how often a wave executes the phase's code per WAVE-ROUND (a wave's
share of one 8192-position round of k_lz77: 512 positions), from the loop structure and the kernel's own stamps (reconcile
rounds, steps per round ...).  Instructions inlined from helper functions above the kernel (rank_of, cmp8 ...) carry the
helpers' line numbers: they inherit the phase of the instruction before them.

Each opcode is weighted by its measured issue cost in core cycles per wave64 instruction and SIMD
(tools/micro/valu_ops.hip, profiles/r03_valu_ops.log / r06_valu_ops.log: 8 waves per SIMD, independent chains):
the plain VOP1/VOP2 integer ops in their 32-bit encoding issue in 2.3-3.0 cycles, everything in a VOP3 / DPP / SDWA
encoding in 4.1-4.5.  Output: per phase the static instruction counts by class, the dynamic counts per wave-round
(static x trips) and the weighted vector cycles, and the kernel total beside the PMC count it must reproduce
(SQ_INSTS_VALU / waves / rounds).
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-S", "--cuda-device-only"]

# measured issue cost, cycles per wave64 instruction per SIMD (profiles/r03_valu_ops.log, r06_valu_ops.log)
FAST = {"v_and_b32": 2.26, "v_or_b32": 2.30, "v_xor_b32": 2.69, "v_sub_u32": 2.48, "v_subrev_u32": 2.48, "v_lshrrev_b32": 2.26,
        "v_mov_b32": 2.32, "v_not_b32": 2.44, "v_add_u32": 3.03, "v_cndmask_b32": 2.25, "v_bitop3_b32": 2.54}
SLOW = {"v_alignbyte_b32": 4.49, "v_ffbl_b32": 4.21, "v_min3_u32": 4.19, "v_mad_i32_i24": 4.65, "v_bfe_u32": 4.22, "v_lshl_or_b32": 4.23,
        "v_mul_lo_u32": 4.18, "v_mul_u32_u24": 4.22, "v_bcnt_u32_b32": 4.17, "v_lshlrev_b64": 5.31, "v_pk_add_u16": 4.19,
        "v_pk_sub_u16": 4.19, "v_and_or_b32": 4.21, "v_perm_b32": 4.25, "v_dot4_u32_u8": 4.28, "v_add3_u32": 4.19,
        "v_lshlrev_b32": 4.08, "v_min_u32": 4.16, "v_alignbit_b32": 4.19, "v_xad_u32": 4.22, "v_sad_u8": 4.17,
        "v_lshl_add_u32": 4.32, "v_lshl_add_u64": 5.31, "v_add_lshl_u32": 4.32}
DEFAULT_VALU = 4.2


def extra_weights(path):
    """More measured opcodes: lines `name   1.234 ms -> ... = 4.21 cycles ...` of a valu_ops log."""
    out = {}
    try:
        for ln in open(path):
            m = re.match(r"(v_\w+)\s.*=\s*([0-9.]+) cycles", ln)
            # (the lone `v_cndmask_b32 ... vcc` line is a chain on VCC without a writer, 23 cycles: not an issue cost --
            # the PAIR line is: compare 4.15 + select 2.25; encodings with a suffix are classed by their suffix anyway)
            if m and "PAIR" not in ln and float(m.group(2)) < 8 and not re.search(r"_(e64|dpp|sdwa)$", m.group(1)):
                out[m.group(1)] = float(m.group(2))
    except OSError:
        pass
    return out


def classify(op):
    """-> (class, weight): class in valu_fast / valu_slow / salu / lds / vmem / smem / branch / wait / barrier / other"""
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    enc = op[len(base) + 1:] if op != base else ""
    if op.startswith("v_"):
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "valu_lane", DEFAULT_VALU
        if base.startswith("v_cmp"):
            return "valu_slow", 4.15
        if enc in ("", "e32") and base in FAST and not (base == "v_or_b32" and enc == ""):
            return "valu_fast", FAST[base]
        if base == "v_cndmask_b32":
            return "valu_slow", 4.22
        return "valu_slow", SLOW.get(base, DEFAULT_VALU)
    if op.startswith("ds_"):
        return "lds", 0.0
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0.0
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem", 0.0
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch", 0.0
    if op == "s_waitcnt":
        return "wait", 0.0
    if op == "s_barrier":
        return "barrier", 0.0
    if op.startswith("s_"):
        return "salu", 0.0
    return "other", 0.0


def compile_asm(src, extra, out):
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def kernel_body(asm_path, pattern, src_name):
    """lines of the first kernel whose mangled name matches `pattern`: [(kind, text)] with kind in loc / label / inst"""
    rx = re.compile(pattern)
    lines, name, on = [], None, False
    for ln in open(asm_path):
        if not on:
            m = re.match(r"^(_Z\w+):", ln)
            if m and rx.search(m.group(1)):
                on, name = True, m.group(1)
            continue
        s = ln.strip()
        if s.startswith(".loc"):
            p = s.split()
            # a line of another file (a HIP header's min(), a builtin wrapper) says nothing about the phase: 0 = "inherit"
            lines.append(("loc", int(p[2]) if os.path.basename(src_name) in s else 0))
        elif re.match(r"^\.LBB\d+_\d+:", ln):
            d = re.search(r"Depth=(\d+)", ln)
            lines.append(("label", (s.split(":")[0], int(d.group(1)) if d else 0)))
        elif lines and lines[-1][0] == "label" and re.match(r"^\s+;.*This (Inner )?Loop Header: Depth=(\d+)", ln):
            lines[-1] = ("label", (lines[-1][1][0], int(re.search(r"Depth=(\d+)", ln).group(1))))
        elif ln.startswith("\t") and re.match(r"^\t[a-z]", ln) and not s.startswith("."):
            lines.append(("inst", s.split(";")[0].strip()))
            if s.startswith("s_endpgm"):
                break
    if name is None:
        raise SystemExit(f"no kernel matches {pattern!r} in {asm_path}")
    return name, lines


def phase_table(src):
    """[(first_line, name, trips, note)] from the `// @phase` markers, and the line range of functions above the first marker"""
    marks = []
    for i, ln in enumerate(open(src), 1):
        m = re.search(r"//\s*@phase\s+(\S+)(.*)", ln)
        if m:
            rest = m.group(2)
            t = re.search(r"trips=([0-9.]+)", rest)
            n = re.search(r"note=(.*)", rest)
            d = re.search(r"depth=(\d+)", rest)
            marks.append((i, m.group(1), float(t.group(1)) if t else 1.0, n.group(1).strip() if n else "", int(d.group(1)) if d else 1))
    return marks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=r"k_lz77ILb0ELb1ELb1ELb1ELb0ELb0ELb0E", help="regex on the mangled kernel name (default: the default effort's k_lz77)")
    ap.add_argument("--src", default=os.path.join(ROOT, "starflate_amd", "csrc", "sf_kernels.hip"))
    ap.add_argument("--json", default=None)
    ap.add_argument("--blocks", action="store_true", help="list every phase's opcodes")
    ap.add_argument("--annotate", default=None, help="write the kernel's assembly with the phase of every instruction to this file")
    ap.add_argument("--pmc-valu-per-wave-round", type=float, default=None, help="SQ_INSTS_VALU / (waves x rounds) of a PMC pass, printed beside the model's total")
    ap.add_argument("--weights-log", default=os.path.join(ROOT, "profiles", "r06_valu_ops.log"))
    args = ap.parse_args()

    for k, v in extra_weights(args.weights_log).items():
        (FAST if k in FAST else SLOW)[k] = v
    marks = phase_table(args.src)
    if not marks:
        raise SystemExit("no `// @phase` markers in the source")
    first_mark = marks[0][0]
    with tempfile.TemporaryDirectory() as td:
        g, p = os.path.join(td, "g.s"), os.path.join(td, "p.s")
        compile_asm(args.src, ["-gline-tables-only"], g)
        compile_asm(args.src, [], p)
        name, body = kernel_body(g, args.kernel, args.src)
        _, plain = kernel_body(p, args.kernel, args.src)
    n_g = sum(1 for k, _ in body if k == "inst")
    n_p = sum(1 for k, _ in plain if k == "inst")

    def phase_of(line):
        cur = None
        for m in marks:
            if m[0] <= line:
                cur = m
            else:
                break
        return cur

    def new_stat(trips, note, depth):
        return {"trips": trips, "note": note, "depth": depth, "static": collections.Counter(), "dyn": collections.Counter(), "ops": collections.Counter()}

    stats = collections.OrderedDict()
    meta = {}
    for m in marks:
        if m[1] != "inherit" and not m[1].startswith("+"):
            stats.setdefault(m[1], new_stat(m[2], m[3], m[4]))
            meta[m[1]] = m
    stats.setdefault("(prologue/epilogue)", new_stat(0.0, "outside the round loop", 0))
    # `inherit`: code of a lambda / helper defined here runs where it is inlined -- it keeps the phase of the instruction before
    # it; `+name trips=f`: a conditional part of such code, a sub-phase `<phase>.<name>` executed f times per execution of <phase>.
    # A phase's `depth` is the loop depth its code belongs to (1: the round loop, 2: a loop inside a round); an instruction the
    # compiler hoisted into a shallower block runs once per round (depth 1) or once per kernel (depth 0: prologue).
    base, sub, depth = None, None, 0
    ann = open(args.annotate, "w") if args.annotate else None
    cur_line = 0
    for kind, v in body:
        if kind == "label":
            depth = v[1]
            if ann:
                ann.write(f"{v[0]}:   ; depth {depth}\n")
            continue
        if kind == "loc":
            cur_line = v
            if v >= first_mark:  # a line of the kernel body; helper-function lines (above the first marker) and line 0 inherit
                m = phase_of(v)
                if m[1] == "inherit":
                    sub = None
                elif m[1].startswith("+"):
                    sub = m
                else:
                    base, sub = m[1], None
            continue
        op = v.split()[0]
        cls, w = classify(op)
        key, trips = "(prologue/epilogue)", 0.0
        if base and depth >= 1:
            key, trips = base, (meta[base][2] if depth >= meta[base][4] else 1.0)
            if sub is not None:
                key = base + "." + sub[1][1:]
                trips *= sub[2]
                if key not in stats:
                    stats[key] = new_stat(meta[base][2] * sub[2], sub[3], meta[base][4])
        st = stats[key]
        if ann:
            ann.write(f"  {key:<24} x{trips:<5g} L{cur_line:<5} {v}\n")
        st["static"][cls] += 1
        st["ops"][op] += 1
        st["dyn"][cls] += trips
        if cls.startswith("valu"):
            st["dyn"]["cycles"] += w * trips
            st["dyn"]["cyc_" + ("fast" if cls == "valu_fast" else "slow")] += w * trips

    hdr = (f"{'phase':<26}{'trips':>6} | {'VALU':>5}{'fast':>5}{'slow':>5}{'lane':>5} {'SALU':>5}{'LDS':>4}{'VMEM':>5}{'br':>4}{'wait':>5} | "
           f"{'dyn VALU':>9}{'dyn cyc':>9}{'cyc/VALU':>9}{'dyn SALU':>9}{'dyn LDS':>8}")
    print(f"kernel {name}\ninstructions: {n_g} (with line tables) / {n_p} (plain build){'' if n_g == n_p else '  ** DIFFER **'}\n")
    print(hdr)
    print("-" * len(hdr))
    tot = collections.Counter()
    rows = []
    for key, st in stats.items():
        s, d = st["static"], st["dyn"]
        valu = s["valu_fast"] + s["valu_slow"] + s["valu_lane"]
        if not sum(s.values()):
            continue
        dyn_valu = d["valu_fast"] + d["valu_slow"] + d["valu_lane"]
        row = {"phase": key, "trips": st["trips"], "static_valu": valu, "static_fast": s["valu_fast"], "static_slow": s["valu_slow"], "static_lane": s["valu_lane"],
               "static_salu": s["salu"], "static_lds": s["lds"], "static_vmem": s["vmem"], "static_branch": s["branch"], "static_wait": s["wait"] + s["barrier"],
               "dyn_valu": round(dyn_valu, 1), "dyn_valu_fast": round(d["valu_fast"], 1), "dyn_valu_cycles": round(d["cycles"], 1),
               "dyn_salu": round(d["salu"] + d["branch"], 1), "dyn_lds": round(d["lds"], 1), "dyn_vmem": round(d["vmem"], 1), "note": st["note"]}
        rows.append(row)
        print(f"{key:<26}{st['trips']:>6.2f} | {valu:>5}{s['valu_fast']:>5}{s['valu_slow']:>5}{s['valu_lane']:>5} {s['salu']:>5}{s['lds']:>4}{s['vmem']:>5}{s['branch']:>4}{s['wait'] + s['barrier']:>5} | "
              f"{dyn_valu:>9.1f}{d['cycles']:>9.1f}{(d['cycles'] / dyn_valu if dyn_valu else 0):>9.2f}{d['salu'] + d['branch']:>9.1f}{d['lds']:>8.1f}")
        for k in ("dyn_valu", "dyn_valu_fast", "dyn_valu_cycles", "dyn_salu", "dyn_lds", "dyn_vmem"):
            tot[k] += row[k]
        if args.blocks:
            print("      " + ", ".join(f"{o} x{c}" for o, c in st["ops"].most_common()))
    print("-" * len(hdr))
    print(f"per wave-round: {tot['dyn_valu']:.0f} vector instructions ({tot['dyn_valu_fast']:.0f} of them in the fast class) = {tot['dyn_valu_cycles']:.0f} weighted issue cycles "
          f"({tot['dyn_valu_cycles'] / max(tot['dyn_valu'], 1):.2f} per instruction), {tot['dyn_salu']:.0f} scalar, {tot['dyn_lds']:.0f} LDS, {tot['dyn_vmem']:.0f} global")
    if args.pmc_valu_per_wave_round:
        print(f"PMC: {args.pmc_valu_per_wave_round:.0f} vector instructions per wave-round (model / PMC = {tot['dyn_valu'] / args.pmc_valu_per_wave_round:.3f})")
    if args.json:
        json.dump({"kernel": name, "instructions": n_p, "rows": rows, "total": dict(tot), "weights": {"fast": FAST, "slow": SLOW, "default": DEFAULT_VALU}},
                  open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
