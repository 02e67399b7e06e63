"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for an MFMA result read by a memory instruction too soon.

CDNA3 ISA, "manually inserted wait states": an XDL (MFMA) write of a VGPR / AGPR followed by a VMEM / LDS / FLAT read of that register needs up to 18 wait states
(16-pass instructions; 10 for the 8-pass v_mfma_f32_16x16x4_f32 / 16x16x32_bf16); the compiler inserts them -- but polyd_edge_kernel showed
`ds_write_b128 v75, a[0:3]` two instructions behind the last `v_mfma_f32_16x16x4_f32 a[0:3]` of the PREVIOUS basic block (reached through an s_branch): the
accumulator's third component came out wrong, differently from run to run, in builds whose code alignment differed.  From every MFMA this walks the control flow
(fall-through, s_branch, both sides of s_cbranch_*) for NEED wait states (s_nop N counts N + 1) and reports stores that read the MFMA's destination inside that window.
Usage: python scripts/mfma_hazard_scan.py file.s [...]     (exit status 1 when something is found)"""
import re, sys
NEED = 11
# wait states the ISA asks for between the MFMA and a VMEM / LDS read of its result: 8-pass v_mfma_f32_16x16x4_f32 (32 cycles) 10, 4-pass v_mfma_f32_16x16x32_bf16 (16 cycles) 6
REQUIRED = {"v_mfma_f32_16x16x4_f32": 10, "v_mfma_f32_16x16x32_bf16": 6}
mf = re.compile(r'v_mfma_\S+\s+([av])\[(\d+):(\d+)\]')
st = re.compile(r'(ds_write\S*|ds_store\S*|global_store\S*|buffer_store\S*|flat_store\S*|scratch_store\S*)\s+(.*)')
reg = re.compile(r'([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b')
def regs(s):
    out = set()
    for m in reg.finditer(s):
        if m.group(1): out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else: out.add((m.group(4), int(m.group(5))))
    return out
total = 0
for path in sys.argv[1:]:
    kernels, cur = [], None
    for ln, l in enumerate(open(path), 1):
        t = l.split(';')[0].strip()
        if not t: continue
        if t.endswith(':'):
            if not t.startswith('.L'):
                cur = {"name": t[:-1], "ins": [], "labels": {}}; kernels.append(cur)
            elif cur is not None: cur["labels"][t[:-1]] = len(cur["ins"])
            continue
        if t.startswith('.') or cur is None: continue
        cur["ins"].append((ln, t))
    hits = 0
    for k in kernels:
        ins, labels = k["ins"], k["labels"]
        for i, (ln, t) in enumerate(ins):
            m = mf.match(t)
            if not m: continue
            dst = {(m.group(1), j) for j in range(int(m.group(2)), int(m.group(3)) + 1)}
            seen, work = set(), [(i + 1, 0)]
            while work:
                p, d = work.pop()
                while p < len(ins) and d < NEED and (p, d) not in seen:
                    seen.add((p, d))
                    l2, t2 = ins[p]
                    op = t2.split()[0]
                    s = st.match(t2)
                    if op.startswith('v_') and not op.startswith('v_mfma') and ',' in t2 and dst & regs(t2.split(',', 1)[1]):      # VALU read (v_accvgpr_read, ...): the compiler pads these to exactly the same count
                        if d < REQUIRED.get(t.split()[0], NEED):
                            hits += 1
                            print("%s: %s: line %d `%s` (VALU) reads the result of the MFMA at line %d after %d wait states (%d required)" % (path.split('/')[-1], k["name"][:60], l2, t2[:50], ln, d, REQUIRED.get(t.split()[0], NEED)))
                        break
                    if s and dst & regs(s.group(2)):
                        if d < REQUIRED.get(t.split()[0], NEED):
                            hits += 1
                            print("%s: %s: line %d `%s` reads the result of the MFMA at line %d after %d wait states (%d required)" % (path.split('/')[-1], k["name"][:60], l2, t2[:50], ln, d, REQUIRED.get(t.split()[0], NEED)))
                        break
                    m2 = mf.match(t2)
                    if m2 and dst & {(m2.group(1), j) for j in range(int(m2.group(2)), int(m2.group(3)) + 1)}: break      # re-defined by a later MFMA: that one is walked on its own
                    if op == 's_endpgm': break
                    if op == 's_branch':
                        p = labels.get(t2.split()[1], len(ins)); d += 1; continue
                    if op.startswith('s_cbranch'):
                        tgt = labels.get(t2.split()[1])
                        if tgt is not None: work.append((tgt, d + 1))
                    d += int(t2.split()[1]) + 1 if op == 's_nop' else 1
                    p += 1
    print("%s: %d" % (path.split('/')[-1], hits))
    total += hits
sys.exit(1 if total else 0)
