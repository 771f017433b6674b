"""ISA census of the compositing kernels: compiles csrc/composite.hip to gfx950 assembly (hipcc -S, the flags of
deblurgs_amd/build.py) and counts instructions per class
  * per (Gaussian, quadrant) PASS  = a basic block of the entry loop that contains v_exp_f32,
  * per list ENTRY                 = the rest of the innermost loop (row fetch from LDS, hit tests, zero-fill, the
                                     per-duplicate reduce-scatter and its LDS store),
  * per BATCH of 64 entries        = the enclosing loop minus the entry loop (gather, quadrant culling, LDS staging, stores).
With the per-class issue costs of profiles/valu_peak_<round>.json (tools/valu_rate under rocprofv3 PMC) this gives the
issue-slot-weighted VALU demand per pass / entry / batch that profiles/make_valu.py turns into roofline.valu.

    python tools/isa_census.py [--json out.json]
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = ["valu_plain", "valu_trans", "valu_cmp", "valu_select", "valu_dpp", "valu_permlane", "valu_mov", "valu_readlane",
           "valu_quarter", "valu_pk", "salu", "branch", "lds", "vmem", "smem", "wait_nop"]


def classify(op):
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait_nop"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_permlane"):
        return "valu_permlane"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "valu_readlane"
    if op.startswith(("v_mad_u64", "v_mad_i64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32")):
        return "valu_quarter"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if "_dpp" in op:
        return "valu_dpp"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith("v_cmp"):
        return "valu_cmp"
    if op.startswith("v_cndmask"):
        return "valu_select"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "valu_mov"
    if op.startswith("v_"):
        return "valu_plain"
    return None


def parse_kernels(asm):
    """-> {kernel name: [(label, loop_depth, {class: count}, has_exp)] per basic block}"""
    kernels, cur, blocks, depth = {}, None, None, 0
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, blocks = m.group(1), []
            kernels[cur] = blocks
            blocks.append(["entry", 0, {}, False])
            continue
        if cur is None:
            continue
        if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m or line.startswith("; %bb."):
            d = re.search(r"Depth=(\d+)", line)
            if d:
                depth = int(d.group(1))
            elif "in Loop" not in line and "Loop Header" not in line and m:
                depth = 0
            blocks.append([m.group(1) if m else line.strip(), depth, {}, False])
            continue
        d = re.search(r"Loop Header: Depth=(\d+)|Parent Loop .* Depth=(\d+)|Inner Loop Header: Depth=(\d+)", line)
        if d and blocks and not blocks[-1][2]:
            vals = [int(x) for x in d.groups() if x]
            if "Inner Loop Header" in line or "This Loop Header" in line:
                blocks[-1][1] = vals[-1]
                depth = vals[-1]
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        if c is None:
            continue
        blocks[-1][2][c] = blocks[-1][2].get(c, 0) + 1
        if op.startswith("v_exp_f32"):
            blocks[-1][3] = True
    return kernels


def census(asm):
    out = {}
    for name, blocks in parse_kernels(asm).items():
        if "composite" not in name:
            continue
        short = "composite_fwd" if "fwd" in name else "composite_bwd"
        m = re.search(r"kernelILb(\d)(?:ELb(\d))?(?:ELb(\d))?", name)
        if m:
            flags = [x for x in m.groups() if x is not None]
            if "fwd" in name and len(flags) == 3:
                if flags[2] == "1":
                    continue           # the parity tests' checksum variant of the forward: not a product kernel
                flags = flags[:2]      # <WITHDEPTH, KEEP>
            short += "<" + ",".join("true" if x == "1" else "false" for x in flags) + ">"
        maxd = max(b[1] for b in blocks)
        passes = [b for b in blocks if b[3] and b[1] == maxd]
        add = lambda bs: {c: sum(b[2].get(c, 0) for b in bs) for c in CLASSES}
        inner = [b for b in blocks if b[1] == maxd]
        outer = [b for b in blocks if b[1] == maxd - 1]
        npass = max(len(passes), 1)
        per_pass = {c: round(v / npass, 2) for c, v in add(passes).items()}
        entry = add([b for b in inner if not b[3]])
        # the entry loop exists once per variant of the batch loop (with / without the power compare): per copy
        copies = max(npass // 4, 1)
        out[short] = {"pass_blocks": npass, "loop_copies": copies,
                      "per_pass": per_pass,
                      "per_entry_outside_passes": {c: round(v / copies, 2) for c, v in entry.items()},
                      "per_batch_outside_entry_loop": add(outer),
                      "static_total": add(blocks)}
    return out


def main():
    sys.path.insert(0, ROOT)
    from deblurgs_amd import build as b
    src = os.path.join(b.CSRC, "composite.hip")
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "composite.s")
        cmd = [b._hipcc()] + [f for f in b.COMMON if f != "-fPIC"] + b.SOURCES["composite.hip"] + ["-S", "--cuda-device-only",
                                                                                                  "-o", s, src]
        subprocess.run(cmd, check=True, capture_output=True)
        res = census(open(s).read())
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    for k, v in res.items():
        print(k)
        for part in ("per_pass", "per_entry_outside_passes", "per_batch_outside_entry_loop"):
            nz = {c: n for c, n in v[part].items() if n}
            valu = sum(n for c, n in nz.items() if c.startswith("valu"))
            print(f"   {part:30s} VALU {valu:6.1f}  {nz}")


if __name__ == "__main__":
    main()
