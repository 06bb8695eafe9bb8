#!/usr/bin/env python3
"""Generates spconv_st_loop.inc: the hand-scheduled offset loop of the staged sparse convolution (spconv_st.hip) as
gfx950 inline-assembly text, one blob per (pass, 32-channel block) of a tile.

Why generated assembly: the C++ version of this loop (27 offsets x NH half-steps, fully unrolled) came out of hipcc at
256 VGPRs with spills, its LDS operand reads issued BEHIND the MFMAs that free their registers (each half-step then
starts with an exposed `s_waitcnt lgkmcnt(0)`), and any scalar branch around an MFMA group made the compiler drain its
load counters at the join.  Here every register is assigned by hand (192 VGPRs), the operand reads of half-step s+1 are
issued ahead of the MFMAs of half-step s (true double buffer), weight fragments run WD offsets ahead and rulebook
entries two, every wait is an exact in-order count, and an EMPTY (16-row chunk, offset) block - no row of the chunk has
a neighbour at the offset, one bit of a mask the rulebook builder stores - costs two scalar instructions and a short
forward branch instead of 6 MFMAs (scripts/micro/mfma_dep.hip: a skipped group costs ~12 cycles next to active ones,
108 when multiplied; out-of-line stubs cost ~30 more).

Register map (per wave; NH row halves of 64 rows, NC = 4 chunks of 16 rows, NTW = 2 output-channel tiles of 16):
  v[64 ..]   accumulators  ACC(h, c, t) = 64 + ((h*4 + c)*2 + t)*4          (pinned operands of the asm statement)
  v[128:191] operand sets  X(s, c, p)   = 128 + s*32 + (c*2 + p)*4          s: half-step parity, p: 0 hi / 1 lo halves
  v[192:239] weight sets   W(s, t, p)   = 192 + s*16 + (t*2 + p)*4          s = k % (WD+1); lane constants behind them
  v[240:251] rulebook sets L(s, h)      = 240 + (s*NH + h)*2                s = k % 3; uint2 = four 16-bit LDS slots
  v[252:255] address temporaries (two in use)
  s[36:49]   occupancy masks: bit (k&1)*16 + h*4 + c of s[36 + k/2]         (pinned operands)
"""
import sys

K, NC, NTW = 27, 4, 2
LO_REGION = 640 * 64      # the LDS stage: 640 rows x 64 B of hi halves, then the same of lo halves (spconv_st.hip)


def gen(NH, WD, LD=2, skip=True, abl=(), K=K, koff=False, tag="", lazy=False, lowl=False):
    """``lazy`` (round 5): the operand reads of a half-step are issued only for its NON-EMPTY blocks.  The eager schedule reads
    every block's two operand pieces whether or not the block is multiplied - a third of the LDS traffic of a loop whose LDS
    array is as busy as its matrix pipe.  A skipped read makes the number of reads in flight data-dependent, so the exact
    in-order ``lgkmcnt`` counts of the eager schedule are gone: a half-step waits for ALL reads (its own: issued one half-step
    earlier), then issues the next half-step's reads in one burst, then multiplies.  Measured level with the eager schedule on all
    five layer shapes (0.392 vs 0.395, 0.146 vs 0.144, 0.181 vs 0.177, 0.239 vs 0.236, 0.316 vs 0.315 ms; bit-identical outputs):
    a third fewer LDS reads buy nothing - diagnostics builds only (variant 3 of -DEYOC_ST_ABLATIONS).
    ``K``: offsets the blob walks (27 = a stride-1 table).  ``koff``: the weight fragments of offset i start at
    ws0 + s[52 + i] instead of ws0 + i * ks - the class-major transposed kernel (spconv_upc.hip) walks the 1, 2, 4 or 8
    offsets of ONE parity class, which are not equidistant in the packed weights (K <= 8 then)."""
    assert not koff or K <= 8
    assert not lowl or NH == 1
    WD, LD = min(WD, K), min(LD, K)
    NWS, NLS = WD + 1, LD + 1
    ACC = lambda h, c, t: 64 + ((h * NC + c) * NTW + t) * 4
    XS = lambda s, c, p: 128 + s * 32 + (c * 2 + p) * 4
    WS = lambda s, t, p: 192 + s * 16 + (t * 2 + p) * 4
    # ``lowl`` (NH = 1 only, round 6): the rulebook sets and lane constants live in v[96..127] (the second half of the accumulator
    # range, which a 64-row wave does not use), so that v[192:255] holds FOUR weight sets: with 4 chunks per offset an offset lasts
    # ~300 cycles and "two offsets ahead" is less than an L2 round trip - weights run 3 offsets ahead, rulebook entries LD (<= 12)
    LB = 96 if lowl else 192 + NWS * 16
    assert not lowl or (NWS <= 4 and 96 + NLS * NH * 2 + 6 <= 128)
    LS = lambda s, h: LB + (s * NH + h) * 2
    # lane constants, computed in the prologue (no VGPR operands: nothing for the compiler to spill around the blob):
    # GH = (lane >> 4) << 4 (XOR term of the hi piece), WL0 = byte offset of the lane's weight fragments of channel tile 0
    # (tile 1: + 64, in the offset field), LV = (lane & 15) * 8 (+ 4096 per 8 offsets) = offset of the lane's rulebook
    # entries, C4 = a prologue temporary; T = the two address temporaries.  They sit behind the rulebook sets when v[..255]
    # has room, else in v[96..] (NH = 1: half the accumulators) or v[56:63] (clobbered on top; the compiler keeps no value there)
    free = (128 if lowl else 256) - (LB + NLS * NH * 2)
    CR = LB + NLS * NH * 2 if free >= 6 else (96 if NH == 1 else 56)
    GH, WL0, LV, C4 = (f"v{CR + i}" for i in range(4))
    T = [CR + 4, None, CR + 5, None]
    vr = lambda n, w=4: f"v[{n}:{n + w - 1}]"
    out, stubs = [], []
    vmq, lgq = [], []          # issue history of VMEM / LDS operations (tags), oldest first

    def emit(s):
        out.append(s)

    done = [-1]                # VMEM operations up to this index of vmq are known to have landed

    def wait_vm(tag):
        if tag not in vmq:
            return
        idx = len(vmq) - 1 - vmq[::-1].index(tag)
        if idx <= done[0]:
            return
        emit(f"s_waitcnt vmcnt({min(len(vmq) - 1 - idx, 63)})")
        done[0] = idx

    def lg_count(tag):
        if tag not in lgq:
            return 15
        idx = len(lgq) - 1 - lgq[::-1].index(tag)
        return min(len(lgq) - 1 - idx, 15)

    def issue_w(k):
        s = k % NWS
        if "now" in abl:
            return
        if koff:
            emit(f"s_add_u32 %[so], %[ws0], s{52 + k}")
        for t in range(NTW):
            for p in range(2):
                emit(f"buffer_load_dwordx4 {vr(WS(s, t, p))}, {WL0}, %[wr], %[so] offen offset:{p * 1024 + t * 64}")
                vmq.append(("W", k))
        if not koff:
            emit("s_add_u32 %[so], %[so], %[ks]")

    def issue_l(k):
        if "nol" in abl:
            return
        if k % 8 == 0 and k > 0:
            emit(f"v_add_u32 {LV}, 0x1000, {LV}")
        for h in range(NH):
            emit(f"global_load_dwordx2 {vr(LS(k % NLS, h), 2)}, {LV}, %[lb] offset:{(k % 8) * 512 + h * 128}")
            vmq.append(("L", k))

    def addr(k, h, c, t0, t1):
        """LDS address of the hi piece of chunk c's rows at (k, h): the 16-bit rulebook field IS the byte address of the
        row's hi pieces (swizzle included); XOR the lane's piece.  The lo piece sits LO_REGION bytes further (offset field)."""
        reg = LS(k % NLS, h) + (c >> 1)
        if "nov" in abl:
            return
        emit(f"v_xor_b32_sdwa v{t0}, {GH}, v{reg} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_{c & 1}")

    def read(hs, c, p, t):
        if "nox" in abl:
            return
        emit(f"ds_read_b128 {vr(XS(hs & 1, c, p))}, v{t}" + (f" offset:{LO_REGION}" if p else ""))
        lgq.append(("X", hs, c))

    def mfma(h, c, t, term, k, hs):
        a = WS(k % NWS, t, 1 if term == 2 else 0)
        b = XS(hs & 1, c, 1 if term == 1 else 0)
        acc = vr(ACC(h, c, t))
        if "nom" in abl:
            return
        emit(f"v_mfma_f32_16x16x32_f16 {acc}, {vr(a)}, {vr(b)}, {acc}")

    # ---- prologue: first weights / rulebook entries, stage landed, barrier, operands of half-step 0
    emit("s_mov_b32 %[so], %[ws0]")
    # lane = 16 g + m.  Output-channel tiles: MFMA row 4 q + r of tile t is channel 8 q + 4 t + r of the wave's 32, so that a
    # lane's two accumulator tuples (t = 0, 1) are 8 CONSECUTIVE channels (16-byte epilogue accesses).  The packed
    # fragments keep channel 16 nt + m' at lane m' of tile nt: lane m of tile t therefore loads from tile nt = ch >> 4, lane
    # position 16 g + (ch & 15), ch = 8 (m >> 2) + 4 t + (m & 3).
    emit(f"v_mbcnt_lo_u32_b32 {LV}, -1, 0")
    emit(f"v_mbcnt_hi_u32_b32 {LV}, -1, {LV}")
    emit(f"v_and_b32 {GH}, 48, {LV}")                       # 16 g
    emit(f"v_and_b32 {LV}, 15, {LV}")                       # m
    emit(f"v_lshrrev_b32 {WL0}, 2, {LV}")
    emit(f"v_lshlrev_b32 {WL0}, 3, {WL0}")                  # 8 (m >> 2)
    emit(f"v_and_b32 {C4}, 3, {LV}")
    emit(f"v_add_u32 {WL0}, {WL0}, {C4}")                   # ch of tile 0 (0 .. 27); tile 1: + 4 = the next 64 bytes of the same tile nt
    emit(f"v_lshrrev_b32 {C4}, 4, {WL0}")                   # nt
    emit(f"v_and_b32 {WL0}, 15, {WL0}")                     # m'
    emit(f"v_add_u32 {WL0}, {WL0}, {GH}")
    emit(f"v_lshlrev_b32 {WL0}, 4, {WL0}")
    emit(f"v_mul_lo_u32 {C4}, {C4}, %[w1]")
    emit(f"v_add_u32 {WL0}, {WL0}, {C4}")
    emit(f"v_lshlrev_b32 {LV}, 3, {LV}")
    for k in range(WD):
        issue_w(k)
    for k in range(LD):
        issue_l(k)
    emit("s_waitcnt vmcnt(0)")
    done[0] = len(vmq) - 1
    emit("s_barrier")
    if "trace" in abl:          # diagnostics build: when did the stage land and the barrier open
        emit("s_memrealtime %[tb]")       # the constant 100 MHz counter (s_memtime's rate follows the clock: round 5's trace had no usable tick)
    for c in range(NC):
        t0, t1 = T[(c & 1) * 2], T[(c & 1) * 2 + 1]
        addr(0, 0, c, t0, t1)
        read(0, c, 0, t0)
        read(0, c, 1, t0)

    # ---- the half-steps
    for k in range(K):
        for h in range(NH):
            hs = k * NH + h
            has_next = hs + 1 < K * NH
            kn, hn = divmod(hs + 1, NH)
            if h == 0 and "trace" in abl and k in (9, 18):
                emit(f"s_memrealtime %[tb{k // 9}]")        # diagnostics build: when the loop reached offsets 9 and 18
            if h == 0:
                if k + WD < K:
                    issue_w(k + WD)
                if k + LD < K:
                    issue_l(k + LD)
                wait_vm(("W", k))
            if has_next:
                wait_vm(("L", kn))
            if lazy:
                emit("s_waitcnt lgkmcnt(0)")
                if has_next:
                    for c in range(NC):
                        t0 = T[(c & 1) * 2]
                        lab = f"{tag}r{kn}h{hn}c{c}"
                        emit(f"s_bitcmp1_b32 s{36 + (kn >> 1)}, {(kn & 1) * 16 + hn * 4 + c}")
                        emit(f"s_cbranch_scc0 .Lst%=_{lab}")
                        addr(kn, hn, c, t0, None)
                        read(hs + 1, c, 0, t0)
                        read(hs + 1, c, 1, t0)
                        emit(f".Lst%=_{lab}:")
                for c in range(NC):
                    lab = f"{tag}k{k}h{h}c{c}"
                    emit(f"s_bitcmp1_b32 s{36 + (k >> 1)}, {(k & 1) * 16 + h * 4 + c}")
                    emit(f"s_cbranch_scc0 .Lst%=_{lab}")
                    for term in range(3):
                        mfma(h, c, 0, term, k, hs)
                        mfma(h, c, 1, term, k, hs)
                    emit(f".Lst%=_{lab}:")
                continue
            for c in range(NC):
                t0, t1 = T[(c & 1) * 2], T[(c & 1) * 2 + 1]
                if has_next:
                    addr(kn, hn, c, t0, t1)
                lab = f"{tag}k{k}h{h}c{c}"
                if has_next:
                    read(hs + 1, c, 0, t0)
                    read(hs + 1, c, 1, t0)
                if skip:
                    emit(f"s_bitcmp1_b32 s{36 + (k >> 1)}, {(k & 1) * 16 + h * 4 + c}")
                    emit(f"s_cbranch_scc0 .Lst%=_{lab}")
                emit(f"s_waitcnt lgkmcnt({lg_count(('X', hs, c))})")
                for term in range(3):
                    mfma(h, c, 0, term, k, hs)
                    mfma(h, c, 1, term, k, hs)
                if skip:
                    emit(f".Lst%=_{lab}:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_nop 7")
    emit("s_nop 7")
    emit("s_nop 7")
    return out


WD2, LD2 = 1, 2            # NH = 2 (measured: weights two offsets ahead and rulebook entries three, lane constants in v[56:61], gain nothing)


def gen_upc(abl=(), lazy=False):
    upc = []
    for kc in (8, 4, 2):
        upc += [f"s_cmp_eq_u32 %[nk], {kc}", f"s_cbranch_scc1 .Lst%=_upc{kc}"]
    for kc in (1, 2, 4, 8):
        if kc > 1:
            upc.append(f".Lst%=_upc{kc}:")
        upc += gen(2, WD2, LD2, True, abl, K=kc, koff=True, tag=f"u{kc}", lazy=lazy)
        if kc < 8:
            upc.append("s_branch .Lst%=_upcend")
    upc.append(".Lst%=_upcend:")
    return upc


def clobbers(lo=128):
    return ", ".join(f'"v{i}"' for i in range(lo, 256))



def write_blob(f, name, lines):
    f.write(f"#define EYOC_ST_LOOP_{name} \\\n")
    for ln in lines:
        f.write(f'  "{ln}\\n\\t" \\\n')
    f.write('  ""\n')


def main(path):
    """spconv_st_loop.inc (committed): the loops the library runs.  spconv_st_loop_abl.inc (not committed; the Makefile writes
    it for -DEYOC_ST_ABLATIONS / -DEYOC_ST_TRACE builds): timing-only ablations, the trace build and measured-not-kept schedules."""
    with open(path, "w") as f:
        f.write("// GENERATED by gen_st_loop.py - do not edit.  The offset loop of spconv_st_kernel as gfx950 assembly text\n")
        f.write("// (register map, schedule and wait counts: see the generator).\n")
        write_blob(f, "NH2", gen(2, WD2, LD2))
        write_blob(f, "NH1", gen(1, 2, 2))
        write_blob(f, "NH1_DEEP", gen(1, 3, 5, lowl=True))      # clobbers v[96:127] on top (EYOC_ST_LOOP_CLOBBERS_NH1)
        write_blob(f, "NH1_LAZY", gen(1, 2, 2, lazy=True))      # operand reads for non-empty blocks only (the NH = 1 loops are LDS-read bound)
        write_blob(f, "NH1_DEEP_LAZY", gen(1, 3, 5, lowl=True, lazy=True))
        write_blob(f, "NH2_NOSKIP", gen(2, WD2, LD2, skip=False))
        # spconv_upc.hip: the offsets of ONE parity class of a transposed (stride 2) table - 1, 2, 4 or 8 of them, %[nk] says how
        # many.  One statement with a scalar dispatch in front (four statements in an if-chain made the compiler shuffle the
        # pinned accumulators through scratch: 168 spilled VGPRs)
        write_blob(f, "UPC", gen_upc())
        f.write(f"#define EYOC_ST_LOOP_CLOBBERS {clobbers()}\n")
        f.write("#define EYOC_ST_LOOP_CLOBBERS_NH1 EYOC_ST_LOOP_CLOBBERS, " + ", ".join(f'"v{i}"' for i in range(96, 128)) + "\n")
    with open(path.replace(".inc", "_abl.inc"), "w") as f:
        f.write("// GENERATED by gen_st_loop.py - diagnostics builds only (EYOC_ST_ABLATIONS / EYOC_ST_TRACE); results are garbage\n")
        for name, abl in (("NOW", ("now",)), ("NOX", ("nox", "nov")), ("NOV", ("nov",)), ("NOM", ("nom",)), ("NOMW", ("nom", "now")),
                          ("NOL", ("nol",)), ("NOWL", ("now", "nol")), ("NOMWL", ("nom", "now", "nol")),
                          ("EMPTY", ("nom", "now", "nol", "nox", "nov")), ("TRACE", ("trace",))):
            write_blob(f, "NH2_" + name, gen(2, WD2, LD2, True, abl))
        write_blob(f, "UPC_NOM", gen_upc(("nom",)))
        write_blob(f, "UPC_NOW", gen_upc(("now",)))
        write_blob(f, "UPC_EMPTY", gen_upc(("nom", "now", "nol", "nox", "nov")))
        write_blob(f, "NH1_EMPTY", gen(1, 2, 2, True, ("nom", "now", "nol", "nox", "nov")))   # round 6: what a 128-row strided tile costs around its loop
        write_blob(f, "NH1_NOM", gen(1, 2, 2, True, ("nom",)))
        write_blob(f, "NH1_NOMW", gen(1, 2, 2, True, ("nom", "now")))
        write_blob(f, "NH1_NOMX", gen(1, 2, 2, True, ("nom", "nox", "nov")))
        write_blob(f, "NH1_NOML", gen(1, 2, 2, True, ("nom", "nol")))
        write_blob(f, "NH2_W2L3", gen(2, 2, 3))       # weights two offsets ahead, rulebook entries three: no gain
        write_blob(f, "NH2_LAZY", gen(2, WD2, LD2, lazy=True))   # operand reads only for non-empty blocks (round 5): level on every layer
        f.write("#define EYOC_ST_LOOP_CLOBBERS_LOW EYOC_ST_LOOP_CLOBBERS, " + ", ".join(f'"v{i}"' for i in range(56, 62)) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else __file__.replace("gen_st_loop.py", "spconv_st_loop.inc"))
