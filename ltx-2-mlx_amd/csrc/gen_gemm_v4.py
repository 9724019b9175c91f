#!/usr/bin/env python3
"""Generator of the hand-scheduled K loops of the 4-wave bf16 GEMM (gemm_v4.hip) for gfx950.

Writes gemm_v4_loop.inc: one C string literal per variant holding the whole K loop as ONE asm block
(prologue DMA, software-pipelined loop, drain).  hipcc never sees inside it: every register is named
here, every wait count is derived here, and `check()` replays the instruction stream against a model
of the LDS stages / fragment register sets to prove the RAW / WAR ordering before anything runs.

Geometry: tile BM x 256 x 64 (BM = 256 or 224), 4 waves = one per SIMD, wave grid WR x WC (1x4 or 2x2),
MFMA block MB = 32 (v_mfma_f32_32x32x16_bf16, 4 k-steps per K-tile) or 16 (v_mfma_f32_16x16x32_bf16,
2 k-steps), C^T orientation (A operand = weight fragment, B operand = activation fragment).  A wave
owns rbw x cbw blocks -> rbw*cbw accumulators in AGPRs.  LDS: [A stage0 32K][A stage1 32K][W stage0 32K]
[W stage1 32K], rows of 128 B (64 bf16), 16-byte chunks XOR-swizzled with (row>>1)&7 on the DMA source.

Per K-tile and wave: NKS k-steps of rbw*cbw MFMAs; the rbw+cbw fragment reads of k-step q+1 are issued
in MFMA slots of k-step q into the other fragment register set; the LDS-DMA pieces of tile t+2 are issued
in the slots of (t, last k-step) and (t+1, ks0); ONE s_barrier per K-tile, before the last k-step:

  (t,ks0) .. (t,ksN-2) | lgkmcnt(0) vmcnt(0) s_barrier | (t,ksN-1) (t+1,ks0) ...
     reads of stage s=t&1 all retired before the barrier -> DMA(t+2) may overwrite stage s after it;
     own pieces of tile t+1 retired before the barrier -> every wave may read stage 1-s after it.

Fixed registers (the .hip passes them with physical-register constraints):
  v[0:15]  voffA[j]  byte offset of this lane's 16 B in activation piece j (row clamp + source swizzle applied)
  v[16:23] voffW[j]
  v[24:27] addrA     LDS byte address of this lane's activation fragment chunk (block 0 of the wave):
                     32x32 blocks: [ks] for ks = 0..3 in stage 0 (the stage is an immediate offset);
                     16x16 blocks: [ks + 2*stage] for ks = 0..1 (per-stage bases: immediate offsets stay < 64 KiB)
  v[28:31] addrW     same for the weight fragment
  v32      conv only: lane i holds the byte offset of tap i in the padded activation volume
  v[36:..] fragment set P, then set Q  (rbw activation fragments, then cbw weight fragments, 4 VGPRs each)
  a[(rb*cbw+cb)*ACC ...] accumulators (ACC = 16 for 32x32 blocks, 4 for 16x16)
Operands: %[ra] %[rw] buffer resources (SGPR quads), %[nk] K-tiles, %[la] %[lw] LDS byte base of this
wave's activation / weight DMA pieces in stage 0, %[kb] first K-tile of this block (split-K; 0 otherwise); conv: %[lcpt] log2(K-tiles per tap), %[cptm1] K-tiles per tap - 1.

fp8-resident weights (w8=True, 16x16x32 layouts): the weight operand is e4m3fn codes [N][K] (1 byte each) plus one fp32 scale per
output column.  The W stage holds 64-byte rows (8-byte chunks XOR-swizzled with 2*((row>>2)&3)); a raw fragment is one
ds_read_b64 (8 codes) into v[RAW..]; it is expanded in the MFMA slots of the SAME k-step into the 4-register bf16 fragment of
the other set: bf16(f32(code) * scale) = v_cvt_pk_f32_fp8 (exact) -> v_pk_mul_f32 (scale pair v[SCL + 2 cb : +1]) ->
v_cvt_pk_bf16_f32 (RNE) -- the arithmetic of the dequantise-at-load kernel, so the GEMM output is bit-identical to it.

Implicit-GEMM conv (conv=True): the activation operand is a PADDED channels-last volume, so tap (kt, kh, kw) of every
output row is the row's own base address plus ONE wave-uniform byte offset; K-tile X covers channels
[64 (X mod cpt), +64) of tap X div cpt.  The activation soffset of tile X is v_readlane(v32, X >> lcpt) + ((X & cptm1) << 7)
(read one k-step before its first use), the weight soffset stays X << 7 (weights are [Cout][taps][Cin], K-contiguous).
"""
import sys

VOFF_A, VOFF_W, ADDR_A, ADDR_W, V_TAB, P_BASE = 0, 16, 24, 28, 32, 36
S_KA = ("s92", "s93")     # activation K byte offset (soffset) of the tile being DMA'd into stage 0 / 1
S_KW = ("s88", "s89")     # weight K byte offset (conv only; dense: the same register as S_KA)
S_T, S_NK1, S_TMP = "s94", "s95", "s96"
S_TAP, S_C0, S_TAPOFF = "s97", ("s98", "s99"), ("s90", "s91")
SCRATCH_S = ["s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"]
VGPR_TOP_MAX = 224
MFMA_ORDER = next((a.split('=')[1] for a in sys.argv if a.startswith('--order=')), 'rbsnake')     # cbmajor | snake | rbmajor | rbsnake (measured best: one operand register changes per MFMA)


class Gen:
    def __init__(self, rbw, cbw, mb=32, npa=8, npw=8, a_stage=32768, w_base=65536, w_stage=32768, conv=False, w8=False,
                 dma_last=None, dma_ks0=None, reads_every=1, m0_early=False,
                 no_dma=False, no_read=False, no_barrier=False, f8=False, f8_scaled=False):
        self.rbw, self.cbw, self.mb = rbw, cbw, mb
        self.npw, self.a_stage, self.w_base, self.w_stage, self.conv = npw, a_stage, w_base, w_stage, conv
        assert npa <= 16 and npw <= 8
        assert w_base >= 2 * a_stage and w_base + 2 * w_stage <= 160 * 1024
        if mb == 32:
            assert a_stage + 8 * 4096 <= 65536 and w_stage + 8 * 4096 <= 65536
        self.w8 = w8
        # f8: BOTH operands are e4m3fn codes (1 byte each); a K-tile is 128 elements = the same 128-byte rows as 64 bf16, so the LDS
        # image, the swizzle and the DMA schedule are the bf16 loop's; the MFMA is v_mfma_f32_32x32x64_f8f6f4 (fp8 x fp8, 2 k-steps of 64
        # per K-tile) on 8-VGPR fragments = 32 bytes per lane, read as TWO ds_read_b128 (the lane's two adjacent 16-byte chunks)
        self.f8, self.f8_scaled = f8, f8_scaled
        if f8:
            assert mb in (32, 16) and not conv and not w8
        # f8 with 16x16 blocks: v_mfma_f32_16x16x128_f8f6f4 takes a WHOLE 128-element K-tile per instruction (measured +16 % FLOP per joule over
        # the 32x32x64 form on random e4m3 operands, tools/micro/mfma_fp8_power.hip), so a K-tile is ONE k-step and its fragments (rbw + cbw of
        # 8 VGPRs) cannot be double-buffered as a set.  "Rolling" form: the activation fragments have ONE register set -- fragment rb of tile T+1
        # is read into rb's registers as soon as rb's last MFMA of tile T is behind (the order is rb-major) -- and only the cbw weight fragments,
        # which every row block uses, have two sets.
        self.roll = f8 and mb == 16
        self.fr = 8 if f8 else 4                  # VGPRs per fragment
        self.rpf = 2 if f8 else 1                 # ds_read_b128 per fragment
        self.s_kw = S_KW if (conv or w8) else S_KA
        self.kw_shift = 6 if w8 else 7               # weight K-tile = 64 bytes of codes / 128 bytes of bf16
        if w8:
            assert mb == 16 and not conv
        self.nks = 1 if self.roll else (2 if (mb == 16 or f8) else 4)
        self.accsz = 16 if mb == 32 else 4
        self.blk_bytes = mb * 128                 # LDS bytes between consecutive row blocks
        self.npa = npa                            # activation DMA pieces per wave (BM / 32)
        self.NPIECE = npa + npw
        self.nmf = rbw * cbw
        self.nfrag = rbw + cbw
        self.q_base = P_BASE + self.fr * self.nfrag
        self.vgpr_top = self.q_base + self.fr * self.nfrag
        if self.roll:                             # A fragments [P_BASE, +8 rbw), weight set P, weight set Q
            self.wsets = (P_BASE + self.fr * rbw, P_BASE + self.fr * (rbw + cbw))
            self.q_base = self.wsets[1]
            # the LAST row block's fragment has two sets too (tile parity): read into its single set it could only be requested behind the
            # tile's final MFMA, and the lgkmcnt(0) that ends the k-step would wait out a whole LDS latency on every K-tile
            self.alast = (P_BASE + self.fr * (rbw - 1), self.wsets[1] + self.fr * cbw)
            self.apar = 0                             # parity of the last fragment's set that frag() hands out (set by kstep_roll)
            self.vgpr_top = self.alast[1] + self.fr
        if f8 and f8_scaled:                      # one VGPR holding the E8M0 code of 1.0 in every byte (v_mfma_scale_* operands)
            self.one_reg = self.vgpr_top
            self.vgpr_top += 1
        if w8:      # raw code pairs (2 per weight fragment), per-column scale pairs (2 per cb), 8 f32 temporaries
            self.raw_base = self.vgpr_top
            self.scl_base = self.raw_base + 2 * cbw
            self.tmp_base = self.scl_base + 2 * cbw
            self.vgpr_top = self.tmp_base + 8
        assert self.vgpr_top <= VGPR_TOP_MAX, self.vgpr_top
        assert rbw * cbw * self.accsz <= 256
        n = self.NPIECE
        if dma_last is None:
            # one piece every `st` slots; first half in the last k-step of tile t, the rest in ks0 of tile t+1
            st = max(1, (2 * self.nmf) // (n + 1))
            slots = list(range(st - 1, self.nmf, st))
            n1 = min(len(slots), (n + 1) // 2)
            dma_last = slots[:n1]
            rest = n - n1
            dma_ks0 = slots[:rest] if rest <= len(slots) else list(range(rest))
        self.dma_last, self.dma_ks0 = dma_last, dma_ks0
        assert len(dma_last) + len(dma_ks0) == n, (dma_last, dma_ks0, n)
        self.reads_every = reads_every
        self.m0_early = m0_early
        assert self.roll or (self.nfrag * self.rpf - 1) * reads_every < self.nmf * (2 if f8 else 1)
        self.no_dma, self.no_read, self.no_barrier = no_dma, no_read, no_barrier
        self.out = []
        self.trace = []

    def emit(self, s):
        self.out.extend(s.split("\n"))

    def frag(self, setbase, idx):       # idx < rbw: activation fragment rb ; rbw + cb: weight fragment cb
        if self.roll:                   # setbase = one of self.wsets; the activation fragments have a single set (the last one: two, by self.apar)
            if idx == self.rbw - 1:
                return self.alast[self.apar]
            return P_BASE + self.fr * idx if idx < self.rbw else setbase + self.fr * (idx - self.rbw)
        return setbase + self.fr * idx

    def acc(self, rb, cb):
        return (rb * self.cbw + cb) * self.accsz

    def m0_for(self, piece, stage):
        if piece < self.npa:
            return f"s_add_u32 m0, %[la], {stage * self.a_stage + piece * 1024}"
        return f"s_add_u32 m0, %[lw], {stage * self.w_stage + (piece - self.npa) * 1024}"

    def dma(self, piece, stage, tile_tag):
        m0 = self.m0_for(piece, stage)
        if piece < self.npa:
            ins = f"buffer_load_dwordx4 v{VOFF_A + piece}, %[ra], {S_KA[stage]} offen lds"
        else:
            ins = f"buffer_load_dwordx4 v{VOFF_W + piece - self.npa}, %[rw], {self.s_kw[stage]} offen lds"
        self.trace.append(("dma", (piece, stage, tile_tag)))
        if self.no_dma and tile_tag != "prologue":
            return [], None
        return [m0], ins

    def read(self, setbase, idx, stage, ks, tile_tag, half=0):
        f = self.frag(setbase, idx) + 4 * half
        is_a = idx < self.rbw
        blk = (idx if is_a else idx - self.rbw) * self.blk_bytes
        base = ADDR_A if is_a else ADDR_W
        if self.w8 and not is_a:  # 8 fp8 codes of (row lr, k-quarter): 64-byte rows, per-stage base registers
            cb = idx - self.rbw
            r = self.raw_base + 2 * cb
            a = f"ds_read_b64 v[{r}:{r + 1}], v{base + ks + 2 * stage} offset:{cb * 16 * 64}"
        elif self.f8:               # address register [2 ks + half]: chunk (4 ks + 2 k-half-of-the-lane + half) ^ swizzle; stage = immediate
            a = f"ds_read_b128 v[{f}:{f + 3}], v{base + 2 * ks + half} offset:{stage * (self.a_stage if is_a else self.w_stage) + blk}"
        elif self.mb == 16:         # per-stage base registers
            a = f"ds_read_b128 v[{f}:{f + 3}], v{base + ks + 2 * stage} offset:{blk}"
        else:
            a = f"ds_read_b128 v[{f}:{f + 3}], v{base + ks} offset:{stage * (self.a_stage if is_a else self.w_stage) + blk}"
        self.trace.append(("read", ((("AL", self.apar) if idx == self.rbw - 1 else "A") if (self.roll and is_a) else setbase, idx, stage, ks, tile_tag, half)))
        if self.no_read and tile_tag != "prologue":
            return None
        return a

    def expand(self, setbase, cb):
        """12 VALU: raw pair of weight fragment cb -> bf16 fragment (4 VGPRs) of `setbase`."""
        r, t, s, d = self.raw_base + 2 * cb, self.tmp_base, self.scl_base + 2 * cb, self.frag(setbase, self.rbw + cb)
        return [f"v_cvt_pk_f32_fp8_e32 v[{t}:{t + 1}], v{r}",
                f"v_cvt_pk_f32_fp8_sdwa v[{t + 2}:{t + 3}], v{r} src0_sel:WORD_1",
                f"v_cvt_pk_f32_fp8_e32 v[{t + 4}:{t + 5}], v{r + 1}",
                f"v_cvt_pk_f32_fp8_sdwa v[{t + 6}:{t + 7}], v{r + 1} src0_sel:WORD_1",
                f"v_pk_mul_f32 v[{t}:{t + 1}], v[{t}:{t + 1}], v[{s}:{s + 1}]",
                f"v_pk_mul_f32 v[{t + 2}:{t + 3}], v[{t + 2}:{t + 3}], v[{s}:{s + 1}]",
                f"v_pk_mul_f32 v[{t + 4}:{t + 5}], v[{t + 4}:{t + 5}], v[{s}:{s + 1}]",
                f"v_pk_mul_f32 v[{t + 6}:{t + 7}], v[{t + 6}:{t + 7}], v[{s}:{s + 1}]",
                f'v_cvt_pk_" LTX2_DT "_f32 v{d}, v{t}, v{t + 1}',
                f'v_cvt_pk_" LTX2_DT "_f32 v{d + 1}, v{t + 2}, v{t + 3}',
                f'v_cvt_pk_" LTX2_DT "_f32 v{d + 2}, v{t + 4}, v{t + 5}',
                f'v_cvt_pk_" LTX2_DT "_f32 v{d + 3}, v{t + 6}, v{t + 7}']

    def mfma(self, setbase, rb, cb):
        a = self.acc(rb, cb)
        w = self.frag(setbase, self.rbw + cb)
        x = self.frag(setbase, rb)
        self.trace.append(("mfma", (setbase, rb, cb) if not self.roll else (setbase, rb, cb, self.apar)))
        if self.roll:
            return f"v_mfma_f32_16x16x128_f8f6f4 a[{a}:{a + 3}], v[{w}:{w + 7}], v[{x}:{x + 7}], a[{a}:{a + 3}]"
        if self.f8:     # cbsz = blgp = 0: both operands OCP e4m3; the scaled form multiplies by 2^(E8M0 - 127) per 32-element block: 1.0 here
            if self.f8_scaled:
                return (f"v_mfma_scale_f32_32x32x64_f8f6f4 a[{a}:{a + 15}], v[{w}:{w + 7}], v[{x}:{x + 7}], a[{a}:{a + 15}], "
                        f"v{self.one_reg}, v{self.one_reg} op_sel_hi:[0,0,0]")
            return f"v_mfma_f32_32x32x64_f8f6f4 a[{a}:{a + 15}], v[{w}:{w + 7}], v[{x}:{x + 7}], a[{a}:{a + 15}]"
        op = 'v_mfma_f32_32x32x16_" LTX2_DT "' if self.mb == 32 else 'v_mfma_f32_16x16x32_" LTX2_DT "'     # bf16 | f16: the build's operand type
        return f"{op} a[{a}:{a + self.accsz - 1}], v[{w}:{w + 3}], v[{x}:{x + 3}], a[{a}:{a + self.accsz - 1}]"

    def read_order(self):
        if self.w8:             # raw weight codes first: they are expanded while the activation reads are still landing
            return list(range(self.rbw, self.nfrag)) + list(range(self.rbw))
        return [self.rbw] + list(range(self.rbw)) + list(range(self.rbw + 1, self.nfrag))     # W0, A0..A(rbw-1), W1..

    # one k-step: MFMAs on `cur`, reads of (rstage, rks) into `nxt`, DMA pieces in the given slots
    def kstep(self, cur, nxt, rstage, rks, rtag, dma_list, dma_slots, dstage, dtag):
        order = [(rb, cb) for cb in range(self.cbw) for rb in range(self.rbw)]
        if MFMA_ORDER == "snake":         # reverse every other column pass: only one operand register changes between MFMAs
            order = [(rb if cb % 2 == 0 else self.rbw - 1 - rb, cb) for cb in range(self.cbw) for rb in range(self.rbw)]
        elif MFMA_ORDER == "rbmajor":     # activation fragment fixed over cbw consecutive MFMAs
            order = [(rb, cb) for rb in range(self.rbw) for cb in range(self.cbw)]
        elif MFMA_ORDER == "rbsnake":
            order = [(rb, cb if rb % 2 == 0 else self.cbw - 1 - cb) for rb in range(self.rbw) for cb in range(self.cbw)]
        reads = [(r, h) for r in self.read_order() for h in range(self.rpf)]
        if self.f8:         # 2 x nfrag reads over nmf slots: two per slot from the start (an MFMA slot is 64 cycles here)
            read_at = {}
            for i, rh in enumerate(reads):
                read_at.setdefault(i // 2, []).append(rh)
        else:
            read_at = {i * self.reads_every: [rh] for i, rh in enumerate(reads)}
        dma_at = dict(zip(dma_slots, dma_list))
        valu_at = {}
        if self.w8 and not (self.no_read and rtag != "prologue"):
            # after `nwait` activation reads are behind the cbw raw reads, lgkmcnt(nwait) proves the raw codes landed (LGKM
            # returns in order); then one expansion instruction per MFMA slot (two where the k-step would run out of slots)
            nwait = min(8, self.rbw)
            s0 = (self.cbw + nwait - 1) * self.reads_every + 1
            ops = []
            for cb in range(self.cbw):
                ops += self.expand(nxt, cb)
            room = self.nmf - 1 - s0
            assert room >= 8, (room, len(ops))
            k = 0
            for slot in range(s0, self.nmf - 1):
                left = self.nmf - 1 - slot
                if room * 2 >= len(ops):
                    take = 2 if (len(ops) - k) > left else 1
                else:                                       # short last-row-tile variants: up to 4 per slot, spread evenly
                    take = -(-(len(ops) - k) // left)
                valu_at[slot] = ops[k:k + take]
                k += take
                if k >= len(ops):
                    break
            assert k >= len(ops)
            valu_at[s0] = [f"s_waitcnt lgkmcnt({nwait})"] + valu_at[s0]
            self.trace.append(("w8_expand", (nxt, s0)))
        for i, (rb, cb) in enumerate(order):
            mf = self.mfma(cur, rb, cb)
            pre, ins = ([], None)
            if i in dma_at:
                pre, ins = self.dma(dma_at[i], dstage, dtag)
            if self.m0_early and ins and i == 0:
                for s in pre:
                    self.emit(s)
                pre = []
            self.emit(mf)
            r = None
            if i in read_at:
                rr = [self.read(nxt, idx, rstage, rks, rtag, half) for idx, half in read_at[i]]
                rr = [x for x in rr if x]
                r = "\n".join(rr) if rr else None
            if self.m0_early:
                if ins:
                    self.emit(ins)          # M0 was written one slot earlier
                if r:
                    self.emit(r)
                for v in valu_at.get(i, []):
                    self.emit(v)
                nxt_dma = dma_at.get(i + 1)
                if nxt_dma is not None and not (self.no_dma and dtag != "prologue"):
                    self.emit(self.m0_for(nxt_dma, dstage))
            else:
                for s in pre:
                    self.emit(s)
                if r:
                    self.emit(r)
                elif ins:
                    self.emit("s_nop 0")          # SALU write of M0 -> LDS-DMA needs one wait state
                if ins:
                    self.emit(ins)
        self.emit("s_waitcnt lgkmcnt(0)")
        self.trace.append(("lgkm0", None))

    def kstep_roll(self, s):
        """The single k-step of tile T (stage s): MFMAs rb-major on (A regs, weight set s); reads of tile T+1 from stage o = 1 - s:
        the weight fragments into set o in the first slots, activation fragment rb one MFMA behind rb's last use (the last one behind
        wait states: no MFMA follows it); every LDS-DMA piece of tile T+2 -> stage s in the given slots."""
        o = 1 - s
        cur, nxt = self.wsets[s], self.wsets[o]
        order = [(rb, cb if rb % 2 == 0 else self.cbw - 1 - cb) for rb in range(self.rbw) for cb in range(self.cbw)]
        read_at = {}
        for cb in range(self.cbw):                                  # weight fragments of T+1: slots 0 .. cbw-1
            read_at.setdefault(cb, []).extend([(self.rbw + cb, 0), (self.rbw + cb, 1)])
        read_at.setdefault(self.cbw, []).extend([(self.rbw - 1, 0), (self.rbw - 1, 1)])       # the last activation fragment of T+1 -> its OTHER set
        for rb in range(self.rbw - 1):                              # activation fragment rb: after the first MFMA of rb + 1
            read_at.setdefault(self.cbw * (rb + 1) + (1 if rb == 0 else 0), []).extend([(rb, 0), (rb, 1)])
        dma_at = dict(zip(self.dma_last, range(self.NPIECE)))
        for i, (rb, cb) in enumerate(order):
            pre, ins = ([], None)
            if i in dma_at:
                pre, ins = self.dma(dma_at[i], s, "T+2")
            if ins and i == 0:
                for x in pre:
                    self.emit(x)
            self.apar = s                                           # MFMAs of tile T read the last fragment's set s ...
            self.emit(self.mfma(cur, rb, cb))
            if ins:
                self.emit(ins)                                      # M0 was written one slot earlier
            self.apar = o                                           # ... reads of tile T+1 fill set o
            for idx, half in read_at.get(i, []):
                r = self.read(nxt, idx, o, 0, "T+1", half)
                if r:
                    self.emit(r)
            nxt_dma = dma_at.get(i + 1)
            if nxt_dma is not None and not self.no_dma:
                self.emit(self.m0_for(nxt_dma, s))
        self.emit("s_waitcnt lgkmcnt(0)")
        self.trace.append(("lgkm0", None))

    def tile_roll(self, s):
        self.trace.append(("tile", s))
        self.emit("s_waitcnt vmcnt(0)")
        self.trace.append(("vm", 0))
        if not self.no_barrier:
            self.emit("s_barrier")
        self.trace.append(("barrier", None))
        self.koff_commit(s)
        self.kstep_roll(s)

    def koff_prefetch(self, s):
        """conv: tile X = min(T + 2, nk - 1) for the tile T living in stage s; tap offset fetched one k-step early."""
        e = self.emit
        e(f"s_add_u32 {S_TMP}, {S_T}, {2 + s}")
        e(f"s_min_u32 {S_TMP}, {S_TMP}, {S_NK1}")
        e(f"s_add_u32 {S_TMP}, {S_TMP}, %[kb]")          # split-K: this block's first K-tile
        e(f"s_lshl_b32 {S_KW[s]}, {S_TMP}, 7")
        e(f"s_and_b32 {S_C0[s]}, {S_TMP}, %[cptm1]")
        e(f"s_lshl_b32 {S_C0[s]}, {S_C0[s]}, 7")
        e(f"s_lshr_b32 {S_TAP}, {S_TMP}, %[lcpt]")
        e("s_nop 3")                                      # SALU write of the lane select -> v_readlane
        e(f"v_readlane_b32 {S_TAPOFF[s]}, v{V_TAB}, {S_TAP}")

    def koff_commit(self, s):
        e = self.emit
        if self.conv:
            e(f"s_add_u32 {S_KA[s]}, {S_TAPOFF[s]}, {S_C0[s]}")
        else:
            # K offset of tile T+2 (clamped to the last tile: dead data into dead slots, uniform counts)
            e(f"s_add_u32 {S_TMP}, {S_T}, {2 + s}")
            e(f"s_min_u32 {S_TMP}, {S_TMP}, {S_NK1}")
            e(f"s_add_u32 {S_TMP}, {S_TMP}, %[kb]")      # split-K: this block's first K-tile
            e(f"s_lshl_b32 {S_KA[s]}, {S_TMP}, 7")
            if self.w8:
                e(f"s_lshl_b32 {S_KW[s]}, {S_TMP}, 6")

    def tile(self, s):
        """K-tile living in stage s.  tags: 'T' this tile, 'T+1', 'T+2'."""
        n1 = len(self.dma_last)
        p1, p2 = list(range(n1)), list(range(n1, self.NPIECE))
        o = 1 - s
        sets = (P_BASE, self.q_base)
        self.trace.append(("tile", s))
        if self.conv:
            assert not self.dma_ks0, "conv: S_KW/S_C0 of stage s are rewritten at the top of its tile, so every piece goes out in the last k-step"
            self.koff_prefetch(s)
        for ks in range(self.nks - 1):
            cur, nxt = sets[ks & 1], sets[(ks + 1) & 1]
            if ks == 0:         # second part of DMA(T+1) -> stage o
                self.kstep(cur, nxt, s, ks + 1, "T", p2, self.dma_ks0, o, "T+1")
            else:
                self.kstep(cur, nxt, s, ks + 1, "T", [], [], 0, "")
        self.emit("s_waitcnt vmcnt(0)")
        self.trace.append(("vm", 0))
        if not self.no_barrier:
            self.emit("s_barrier")
        self.trace.append(("barrier", None))
        self.koff_commit(s)
        # last k-step: reads (T+1, ks0) from stage o ; first part of DMA(T+2) -> stage s
        ks = self.nks - 1
        self.kstep(sets[ks & 1], sets[(ks + 1) & 1], o, 0, "T+1", p1, self.dma_last, s, "T+2")

    def build(self):
        NP = self.NPIECE
        n1 = len(self.dma_last)
        e = self.emit
        e(f"s_sub_u32 {S_NK1}, %[nk], 1")
        if self.conv:
            # tile X (X = kb, kb + 1): tap X >> lcpt, channels 64 * (X & cptm1)
            e(f"s_lshl_b32 {S_KW[0]}, %[kb], 7")
            e(f"s_and_b32 {S_C0[0]}, %[kb], %[cptm1]")
            e(f"s_lshl_b32 {S_C0[0]}, {S_C0[0]}, 7")
            e(f"s_lshr_b32 {S_TAP}, %[kb], %[lcpt]")
            e("s_nop 3")
            e(f"v_readlane_b32 {S_KA[0]}, v{V_TAB}, {S_TAP}")
            e("s_nop 3")
            e(f"s_add_u32 {S_KA[0]}, {S_KA[0]}, {S_C0[0]}")
            e(f"s_min_u32 {S_TMP}, 1, {S_NK1}")
            e(f"s_add_u32 {S_TMP}, {S_TMP}, %[kb]")
            e(f"s_lshl_b32 {S_KW[1]}, {S_TMP}, 7")
            e(f"s_and_b32 {S_C0[1]}, {S_TMP}, %[cptm1]")
            e(f"s_lshl_b32 {S_C0[1]}, {S_C0[1]}, 7")
            e(f"s_lshr_b32 {S_TAP}, {S_TMP}, %[lcpt]")
            e("s_nop 3")
            e(f"v_readlane_b32 {S_KA[1]}, v{V_TAB}, {S_TAP}")
            e("s_nop 3")                                  # VALU write of an SGPR -> SALU / VMEM read
            e(f"s_add_u32 {S_KA[1]}, {S_KA[1]}, {S_C0[1]}")
            e("s_nop 3")
        else:
            e(f"s_lshl_b32 {S_KA[0]}, %[kb], 7")
            e(f"s_min_u32 {S_TMP}, 1, {S_NK1}")
            e(f"s_add_u32 {S_TMP}, {S_TMP}, %[kb]")
            e(f"s_lshl_b32 {S_KA[1]}, {S_TMP}, 7")
            if self.w8:
                e(f"s_lshl_b32 {S_KW[0]}, %[kb], 6")
                e(f"s_lshl_b32 {S_KW[1]}, {S_TMP}, 6")
        # tile 0 -> stage 0 (all pieces), first part of tile 1 -> stage 1 (rolling form: all of it)
        if self.roll:
            n1 = NP
        for stage, pieces in ((0, range(NP)), (1, range(n1))):
            for p in pieces:
                pre, ins = self.dma(p, stage, "prologue")
                for s in pre:
                    e(s)
                e("s_nop 0")
                e(ins)
        for r in range(self.rbw * self.cbw * self.accsz):
            e(f"v_accvgpr_write_b32 a{r}, 0")
        if self.f8 and self.f8_scaled:
            e(f"v_mov_b32 v{self.one_reg}, 0x7f7f7f7f")
        e(f"s_waitcnt vmcnt({n1})")
        self.trace.append(("vm", n1))
        e("s_barrier")
        self.trace.append(("barrier", None))
        if self.roll:
            self.apar = 0
        for idx in self.read_order():
            for half in range(self.rpf):
                e(self.read(self.wsets[0] if self.roll else P_BASE, idx, 0, 0, "prologue", half))
        e("s_waitcnt lgkmcnt(0)")
        self.trace.append(("lgkm0", None))
        if self.w8:
            for cb in range(self.cbw):
                for v in self.expand(P_BASE, cb):
                    e(v)
        e(f"s_mov_b32 {S_T}, 0")
        e("LTX2_V4_LOOP_%=:")
        self.trace.append(("loop", None))
        if self.roll:
            self.tile_roll(0)
            self.tile_roll(1)
        else:
            self.tile(0)
            self.tile(1)
        e(f"s_add_u32 {S_T}, {S_T}, 2")
        e(f"s_cmp_lt_u32 {S_T}, %[nk]")
        e("s_cbranch_scc1 LTX2_V4_LOOP_%=")
        e("s_waitcnt vmcnt(0)")
        e("s_nop 15")        # XDL write -> v_accvgpr_read of the epilogue: hipcc cannot see the MFMAs in here
        e("s_nop 15")
        return self.out


def check(g, iters=3):
    """Replay the prologue and `iters` loop iterations of the trace with absolute tile numbers and verify:
       RAW(LDS)  a fragment read of tile X is issued only after all of this wave's pieces of X were issued, a counted
                 vmcnt retired them, and an s_barrier followed (the other waves ran the same program up to it);
       WAR(LDS)  a DMA piece of tile X into stage X&1 is issued only after every read of tile X-2 was retired by an
                 lgkmcnt(0) that precedes an s_barrier that precedes the DMA;
       RAW(reg)  an MFMA consumes fragments whose reads were retired, both of the same (tile, ks), and every
                 (tile, ks, rb, cb) product is issued exactly once;
       WAR(reg)  a read overwrites a fragment register only after at least one further MFMA was issued behind the last
                 MFMA that consumed it."""
    tr = g.trace
    li = [i for i, (k, _) in enumerate(tr) if k == "loop"][0]
    body = tr[li + 1:]
    NP, nfrag, nks, rpf = g.NPIECE, g.nfrag, g.nks, g.rpf
    ev = list(tr[:li])
    for it in range(iters):
        base = 2 * it
        cur = None
        for k, pl in body:
            if k == "tile":
                cur = base + pl
                continue
            if k == "dma":
                piece, stage, tag = pl
                ev.append((k, (piece, stage, cur + {"T+1": 1, "T+2": 2}[tag])))
            elif k == "read":
                setb, idx, stage, ks, tag, half = pl
                ev.append((k, (setb, idx, stage, ks, cur + {"T": 0, "T+1": 1}[tag], half)))
            else:
                ev.append((k, pl))
    errors = []
    dma_order = []
    retired_upto = 0
    landed_pos = {}
    barriers = []
    reads_by_tile = {}
    pending_reads = []
    frag = {}
    halves = {}
    frag_last_use = {}
    n_mfma = 0
    done = {}
    stage_tile = {}
    for pos, (k, pl) in enumerate(ev):
        if k == "dma":
            piece, stage, tile = pl
            if isinstance(tile, str):
                tile = stage          # prologue: tile == stage
            if tile % 2 != stage:
                errors.append(f"tile {tile} DMA'd into stage {stage}")
            prev = tile - 2
            if prev >= 0:
                for ip, rp in reads_by_tile.get(prev, []):
                    if rp[0] is None or not any(rp[0] < b < pos for b in barriers):
                        errors.append(f"DMA of tile {tile} piece {piece} at {pos} before reads of tile {prev} retired + barrier")
                        break
                if len(reads_by_tile.get(prev, [])) != nks * nfrag * rpf:
                    errors.append(f"DMA of tile {tile} at {pos}: tile {prev} has only {len(reads_by_tile.get(prev, []))} reads issued so far")
            dma_order.append((pos, tile, piece))
            stage_tile[stage] = tile
        elif k == "vm":
            upto = len(dma_order) - pl
            for i in range(retired_upto, max(retired_upto, upto)):
                landed_pos[(dma_order[i][1], dma_order[i][2])] = pos
            retired_upto = max(retired_upto, upto)
        elif k == "barrier":
            barriers.append(pos)
        elif k == "read":
            setb, idx, stage, ks, tile, half = pl
            if isinstance(tile, str):
                tile = 0
            if tile % 2 != stage or stage_tile.get(stage) != tile:
                errors.append(f"read of tile {tile} from stage {stage} (holds {stage_tile.get(stage)})")
            for p in range(NP):
                lp = landed_pos.get((tile, p))
                if lp is None or not any(lp < b < pos for b in barriers):
                    errors.append(f"read of tile {tile} ks{ks} at {pos}: piece {p} not landed + barrier")
                    break
            if (setb, idx) in frag_last_use and n_mfma - frag_last_use[(setb, idx)] < 1:
                errors.append(f"read into ({setb},{idx}) at {pos} right behind its last consumer")
            frag.pop((setb, idx), None)
            rec = [None]
            reads_by_tile.setdefault(tile, []).append((pos, rec))
            pending_reads.append((setb, idx, tile, ks, rec, half))
        elif k == "lgkm0":
            for setb, idx, tile, ks, rec, half in pending_reads:
                rec[0] = pos
                got = halves.setdefault((setb, idx), {})
                if got.get("of") != (tile, ks):
                    got.clear()
                    got["of"] = (tile, ks)
                got[half] = True
                if all(got.get(h) for h in range(rpf)):       # a fragment is whole once every one of its ds_read_b128 has landed
                    frag[(setb, idx)] = (tile, ks)
            pending_reads = []
        elif k == "mfma":
            setb, rb, cb = pl[:3]
            n_mfma += 1
            seta = (("AL", pl[3]) if rb == g.rbw - 1 else "A") if g.roll else setb      # rolling form: one activation fragment set (the last: two)
            ca, cw = frag.get((seta, rb)), frag.get((setb, g.rbw + cb))
            if ca is None or cw is None:
                errors.append(f"MFMA at {pos} consumes an unretired fragment")
            elif ca != cw:
                errors.append(f"MFMA at {pos} mixes {ca} and {cw}")
            else:
                key = (ca[0], ca[1], rb, cb)
                done[key] = done.get(key, 0) + 1
            frag_last_use[(seta, rb)] = n_mfma
            frag_last_use[(setb, g.rbw + cb)] = n_mfma
        elif k == "nop":
            n_mfma += 1             # wait states behind the last MFMA of a k-step count as one MFMA slot for the WAR(reg) rule
    for t in range(2 * iters):
        for ks in range(nks):
            for rb in range(g.rbw):
                for cb in range(g.cbw):
                    if done.get((t, ks, rb, cb), 0) != 1:
                        errors.append(f"product tile {t} ks{ks} rb{rb} cb{cb} issued {done.get((t, ks, rb, cb), 0)} times")
    return errors


def variant(name, rbw, cbw, **kw):
    g = Gen(rbw, cbw, **kw)
    lines = g.build()
    abl = kw.get("no_dma") or kw.get("no_read") or kw.get("no_barrier")
    errs = check(g)
    if errs and not abl:
        raise SystemExit(f"{name}: pipeline check failed:\n  " + "\n  ".join(errs[:20]))
    body = "".join(f'    "{ln}\\n"\n' for ln in lines)
    extra = ""
    if g.w8:        # per-variant clobber list: the scale registers are inputs
        regs = [i for i in range(33, VGPR_TOP_MAX) if not (g.scl_base <= i < g.scl_base + 2 * g.cbw)]
        extra = (f"#define {name}_SCL \"{{v[{g.scl_base}:{g.scl_base + 2 * g.cbw - 1}]}}\"\n#define {name}_CLOBBERS \\\n    " + ", ".join(f'"v{i}"' for i in regs) +
                 ", \\\n    " + ", ".join(f'"{s}"' for s in SCRATCH_S) + ', "scc", "memory"\n\n')
    hdr = f"// {name}: f8={g.f8} w8={g.w8} rbw={rbw} cbw={cbw} mb={g.mb} npa={g.npa} npw={g.npw} a_stage={g.a_stage} w_base={g.w_base} w_stage={g.w_stage} conv={g.conv} dma_last={g.dma_last} dma_ks0={g.dma_ks0} reads_every={g.reads_every} vgpr_top={g.vgpr_top}"
    return hdr + f"\n#define {name} \\\n" + body.replace('\n', ' \\\n').rstrip(' \\\n') + "\n\n" + extra


def main():
    out = ["// GENERATED by gen_gemm_v4.py -- do not edit.  One asm string per K-loop variant (see the generator's docstring).\n",
           "#pragma once\n\n",
           "#define LTX2_V4_CLOBBERS \\\n    " +
           ", ".join(f'"v{i}"' for i in range(33, VGPR_TOP_MAX)) + ", \\\n    " +
           ", ".join(f'"{s}"' for s in SCRATCH_S) + ', "scc", "memory"\n\n']
    odd16 = list(range(1, 16, 2))
    # (round 3: layouts 0 / 1 / 2 -- 1x4 and 2x2 waves on 32x32x16, 2x2 waves on 16x16x32 -- lost every same-box comparison with layout 3
    # in round 2 and were deleted; the generator's mb = 32 and 2x2 code paths stay: the fp8 variants and the probe harness use them)
    # layout 3: 1x4 waves, 16x16x32: 14|16 x 4 blocks (balanced for 224 rows); dense and conv
    d14, d16 = list(range(0, 56, 4)) + [55], list(range(3, 64, 4))
    for conv in (False, True):
        sfx = "_CONV" if conv else ""
        out.append(variant("LTX2_V4_L14_M16_RB14" + sfx, 14, 4, mb=16, npa=7, dma_last=d14, dma_ks0=[], m0_early=True, conv=conv))
        out.append(variant("LTX2_V4_L14_M16_RB16" + sfx, 16, 4, mb=16, npa=8, dma_last=d16, dma_ks0=[], m0_early=True, conv=conv))
        if not conv:
            # the ragged LAST row tile of a 224-row grid (3456 rows = 15 tiles + 96 rows; 13824 = 61 + 160): only the row blocks
            # that hold real rows are staged and multiplied -- the chip runs at its power cap, so MFMAs on clamped duplicate
            # rows cost real time elsewhere
            out.append(variant("LTX2_V4_L14_M16_RB6", 6, 4, mb=16, npa=3, dma_last=list(range(1, 23, 2)), dma_ks0=[], m0_early=True))
            out.append(variant("LTX2_V4_L14_M16_RB10", 10, 4, mb=16, npa=5, dma_last=list(range(1, 40, 3)), dma_ks0=[], m0_early=True))
        # layout 4: BN = 128, 4x1 waves, 16x16x32: 7|8 x 8 blocks per wave (tile 448|512 x 128)
        out.append(variant("LTX2_V4_L41_M16_RB7" + sfx, 7, 8, mb=16, npa=14, npw=4, a_stage=57344, w_base=114688, w_stage=16384,
                           dma_last=list(range(0, 54, 3)), dma_ks0=[], m0_early=True, conv=conv))
        out.append(variant("LTX2_V4_L41_M16_RB8" + sfx, 8, 8, mb=16, npa=16, npw=4, a_stage=65536, w_base=131072, w_stage=16384,
                           dma_last=list(range(0, 60, 3)), dma_ks0=[], m0_early=True, conv=conv))
        if conv:
            # round 4, layout 7: BN = 64, 4x1 waves, 8 x 4 blocks per wave (tile 512 x 64): the decoder's conv_out (128 -> 48 channels)
            out.append(variant("LTX2_V4_L41N_M16_RB8" + sfx, 8, 4, mb=16, npa=16, npw=2, a_stage=65536, w_base=131072, w_stage=8192,
                               dma_last=list(range(0, 30, 2)) + [29, 30, 31], dma_ks0=[], m0_early=True, conv=conv))
            # a 384-row tile for grids whose 448 / 512-row forms end in a nearly empty round (gemm_v4_conv_launch picks by whole rounds)
            out.append(variant("LTX2_V4_L41_M16_RB6" + sfx, 6, 8, mb=16, npa=12, npw=4, a_stage=49152, w_base=98304, w_stage=16384,
                               dma_last=list(range(0, 48, 3)), dma_ks0=[], m0_early=True, conv=conv))
    # layout 3 with fp8-resident weights: W stage = 256 rows x 64 B
    out.append(variant("LTX2_V4_L14_M16_RB14_W8", 14, 4, mb=16, npa=7, npw=4, w_stage=16384, dma_last=list(range(0, 44, 4)), dma_ks0=[], m0_early=True, w8=True))
    out.append(variant("LTX2_V4_L14_M16_RB16_W8", 16, 4, mb=16, npa=8, npw=4, w_stage=16384, dma_last=list(range(3, 50, 4)), dma_ks0=[], m0_early=True, w8=True))
    out.append(variant("LTX2_V4_L14_M16_RB6_W8", 6, 4, mb=16, npa=3, npw=4, w_stage=16384, dma_last=list(range(1, 22, 3)), dma_ks0=[], m0_early=True, w8=True))
    out.append(variant("LTX2_V4_L14_M16_RB10_W8", 10, 4, mb=16, npa=5, npw=4, w_stage=16384, dma_last=list(range(1, 37, 4)), dma_ks0=[], m0_early=True, w8=True))
    # layout 5: BOTH operands fp8 (e4m3fn codes, K-tile = 128 elements), 1x4 waves, v_mfma_f32_32x32x64_f8f6f4: 7|8 x 2 blocks of 32
    # (the v_mfma_scale_* form with unit block scales -- f8_scaled=True -- measured the same: 166.4 vs 165.4 us on the QKV shape)
    out.append(variant("LTX2_V4_F8_RB7", 7, 2, npa=7, dma_last=[1, 3, 5, 7, 9, 11, 13], dma_ks0=[0, 2, 4, 6, 8, 10, 12, 13], f8=True))
    out.append(variant("LTX2_V4_F8_RB8", 8, 2, npa=8, dma_last=odd16, dma_ks0=odd16, f8=True))
    # layout 6: the same operands on v_mfma_f32_16x16x128_f8f6f4 (one k-step per K-tile, rolling activation fragments), 14 x 4 blocks of 16: 224-row tiles
    out.append(variant("LTX2_V4_F8_M16_RB14", 14, 4, mb=16, npa=7, dma_last=list(range(2, 54, 3))[:15], dma_ks0=[], f8=True))
    if "--probe" in sys.argv:       # ablations of the DiT default for tools/micro/gemm_v4_probe.hip
        out.append(variant("LTX2_V4_L14_M16_RB16_NODMA", 16, 4, mb=16, npa=8, dma_last=d16, dma_ks0=[], m0_early=True, no_dma=True))
        out.append(variant("LTX2_V4_L14_M16_RB16_NOREAD", 16, 4, mb=16, npa=8, dma_last=d16, dma_ks0=[], m0_early=True, no_read=True))
    path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gemm_v4_loop.inc"
    with open(path, "w") as f:
        f.write("".join(out))
    print(f"wrote {path}")


if __name__ == "__main__":
    main()
