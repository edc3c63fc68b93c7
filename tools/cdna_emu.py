#!/usr/bin/env python3
"""A small functional emulator of the gfx950 instruction subset that tools/gen_attn_*.py emit, so a generated kernel body can be
executed on the CPU (one workgroup, its waves run barrier to barrier) and compared with a numpy restatement of the operator:
tests/test_attn_asm_emu_cpu.py.  Test infrastructure -- nothing in orbit-2_amd/ imports it.

What it models: 64-lane VGPR / AGPR files, SGPRs, VCC / EXEC / M0 / SCC, one LDS per workgroup, a flat byte-addressed global
memory, MFMA 32x32x16 bf16 with the hardware's operand / result lane maps, ds_read_b64_tr_b16's 16-lane-group transpose,
LDS-DMA (buffer_load ... lds: M0 base + lane x 16), SDWA byte selects on VOPC.
What it checks besides the values (strict=True), because nothing else can without the hardware:
  * a register that an outstanding LDS / global load will still write is not touched before an s_waitcnt that retires the load
    (vmcnt / lgkmcnt are counted in issue order, as the hardware does);
  * LDS bytes written by an LDS-DMA are read only after the issuing wave's vmcnt wait AND (for other waves) a later barrier;
    an LDS-DMA does not overwrite bytes another wave has read since that wave's last barrier (or has not yet waited for);
  * software wait states the hardware does not interlock: MFMA result -> any non-MFMA reader / overwriting writer (18 issue
    slots), vector write -> MFMA operand (2), transcendental result -> next vector instruction (1), M0 write -> LDS-DMA (1),
    vector write -> v_permlane32_swap (2).
Timing is not modelled."""
import re
import struct

import numpy as np

MFMA_TO_VALU, VALU_TO_MFMA, TRANS_TO_VALU, M0_TO_DMA, VALU_TO_PERM = 18, 2, 1, 1, 2
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32")


class EmuError(Exception):
    pass


def f32(u):
    return np.asarray(u, dtype=np.uint32).view(np.float32)


def u32(f):
    return np.asarray(f, dtype=np.float32).view(np.uint32)


def bf16_round(x):
    """fp32 array -> bf16 bits (uint32 in the low 16), round to nearest even, NaN stays NaN"""
    u = u32(x).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) & 0xFFFF
    nan = np.isnan(np.asarray(x, dtype=np.float32))
    return np.where(nan, np.uint32(0x7FC0), r).astype(np.uint32)


def bf16_to_f32(b):
    return f32((np.asarray(b, dtype=np.uint32) & 0xFFFF) << 16)


class Wave:
    def __init__(self, wg, wid):
        self.wg, self.wid = wg, wid
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.a = np.zeros((256, 64), dtype=np.uint32)
        self.s = np.zeros(128, dtype=np.uint64)        # 32-bit values kept in 64-bit slots
        self.vcc, self.exec, self.m0, self.scc = 0, (1 << 64) - 1, 0, 0
        self.pc, self.done, self.at_barrier = 0, False, False
        self.epoch = 0                                  # barriers passed
        self.issue = 0                                  # issue slots so far (wait-state bookkeeping)
        self.vm, self.lgkm = [], []                     # outstanding ops, oldest first: dict(regs=set, dma=(lo, hi) | None)
        self.wtime = {}                                 # (kind, idx) -> (issue slot, class of the writer)
        self.m0_time = -100
        self.counts = {}


class Workgroup:
    def __init__(self, nwaves, lds_bytes, mem_bytes, strict=True):
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.mem = np.zeros(mem_bytes, dtype=np.uint8)
        self.waves = [Wave(self, w) for w in range(nwaves)]
        self.strict = strict
        # per KiB of LDS: pending DMA (wave, landed epoch or None) and readers since their last barrier
        nk = (lds_bytes + 1023) // 1024
        self.dma_state = [None] * nk                    # None | dict(wave=, retired_epoch=None | int)
        self.readers = [dict() for _ in range(nk)]      # wave id -> epoch of an un-barriered read


_tok = re.compile(r"^(v|a|s)(\d+)$")
_rng = re.compile(r"^(v|a|s)\[(\d+):(\d+)\]$")


def parse_reg(tok):
    m = _tok.match(tok)
    if m:
        return m.group(1), int(m.group(2)), 1
    m = _rng.match(tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
    return None


def parse_program(lines, binds):
    """lines: instruction strings with %[name] operands and %= label suffixes; binds: name -> register text"""
    prog, labels = [], {}
    for raw in lines:
        t = raw.strip().replace("%=", "")
        for k, val in binds.items():
            t = t.replace("%%[%s]" % k, val)
        if "%[" in t:
            raise EmuError("unbound operand in: " + raw)
        if t.endswith(":"):
            labels[t[:-1]] = len(prog)
            continue
        t = re.sub(r"quad_perm:\[(\d),(\d),(\d),(\d)\]", r"quad_perm:\1\2\3\4", t)
        parts = t.split(None, 1)
        op = parts[0]
        ops, mods = [], {}
        if len(parts) > 1:
            fields = [x.strip() for x in parts[1].split(",")]
            last = fields[-1].split()
            if last:
                fields[-1] = last[0]
                for m in last[1:]:
                    if ":" in m:
                        k, val = m.split(":")
                        mods[k] = val
                    else:
                        mods[m] = True
            # a modifier-only tail such as "vmcnt(0) lgkmcnt(0)" stays in fields for s_waitcnt
            ops = fields
            if op == "s_waitcnt":
                ops = parts[1].split()
        for o in ops:                      # scalar register tuples must be aligned (the assembler rejects them otherwise)
            r = parse_reg(o)
            if r and r[0] == "s" and ((r[2] == 2 and r[1] % 2) or (r[2] >= 4 and r[1] % 4)):
                raise EmuError("misaligned scalar register tuple in: " + t)
        prog.append((op, ops, mods, t))
    return prog, labels


class Emu:
    def __init__(self, lines, binds_per_wave, nwaves=4, lds_bytes=160 * 1024, mem_bytes=1 << 24, strict=True):
        self.wg = Workgroup(nwaves, lds_bytes, mem_bytes, strict)
        self.progs = []
        for w in range(nwaves):
            self.progs.append(parse_program(lines, binds_per_wave[w]))
        self.strict = strict

    # ---- operand access -------------------------------------------------------------------------------------------
    def _src(self, w, tok, as_float=False):
        """32-bit source operand -> uint32[64]"""
        r = parse_reg(tok)
        if r:
            kind, idx, n = r
            if kind == "v":
                self._touch(w, ("v", idx), read=True)
                return w.v[idx].copy()
            if kind == "a":
                self._touch(w, ("a", idx), read=True)
                return w.a[idx].copy()
            return np.full(64, np.uint32(int(w.s[idx]) & 0xFFFFFFFF), dtype=np.uint32)
        if tok == "vcc":
            return np.full(64, np.uint32(w.vcc & 0xFFFFFFFF), dtype=np.uint32)
        if tok == "m0":
            return np.full(64, np.uint32(w.m0), dtype=np.uint32)
        return np.full(64, np.uint32(self._imm(tok, as_float)), dtype=np.uint32)

    @staticmethod
    def _imm(tok, as_float=False):
        if tok.startswith("0x") or tok.startswith("-0x"):
            return int(tok, 16) & 0xFFFFFFFF
        if re.match(r"^-?\d+$", tok):
            val = int(tok)
            if as_float and -16 <= val <= 64:
                # integer inline constants are integers even in float instructions (0 is +0.0 either way)
                return val & 0xFFFFFFFF
            return val & 0xFFFFFFFF
        if re.match(r"^-?\d+\.\d*$", tok):
            return struct.unpack("<I", struct.pack("<f", float(tok)))[0]
        raise EmuError("cannot parse operand " + tok)

    def _sval(self, w, tok):
        r = parse_reg(tok)
        if r:
            kind, idx, n = r
            if kind != "s":
                raise EmuError("scalar operand expected: " + tok)
            if n == 2:
                return (int(w.s[idx]) & 0xFFFFFFFF) | ((int(w.s[idx + 1]) & 0xFFFFFFFF) << 32)
            return int(w.s[idx]) & 0xFFFFFFFF
        if tok == "m0":
            return w.m0
        if tok == "exec":
            return w.exec
        if tok == "vcc":
            return w.vcc
        return self._imm(tok)

    def _sset(self, w, tok, val):
        if tok == "exec_lo":
            w.exec = (w.exec & ~0xFFFFFFFF) | (val & 0xFFFFFFFF)
            return
        if tok == "exec_hi":
            w.exec = (w.exec & 0xFFFFFFFF) | ((val & 0xFFFFFFFF) << 32)
            return
        if tok == "m0":
            w.m0 = val & 0xFFFFFFFF
            w.m0_time = w.issue
            return
        if tok == "exec":
            w.exec = val & ((1 << 64) - 1)
            return
        if tok == "vcc":
            w.vcc = val & ((1 << 64) - 1)
            return
        kind, idx, n = parse_reg(tok)
        if n == 2:
            w.s[idx] = val & 0xFFFFFFFF
            w.s[idx + 1] = (val >> 32) & 0xFFFFFFFF
        else:
            w.s[idx] = val & 0xFFFFFFFF

    def _mask(self, w):
        return np.array([(w.exec >> l) & 1 for l in range(64)], dtype=bool)

    def _vset(self, w, tok, val, cls="valu"):
        kind, idx, n = parse_reg(tok)
        file = w.v if kind == "v" else w.a
        m = self._mask(w)
        self._touch(w, (kind, idx), read=False)
        file[idx] = np.where(m, np.asarray(val, dtype=np.uint32), file[idx])
        w.wtime[(kind, idx)] = (w.issue, cls)

    # ---- hazard bookkeeping ---------------------------------------------------------------------------------------------
    def _touch(self, w, reg, read):
        if not self.strict:
            return
        for q, name in ((w.vm, "vmcnt"), (w.lgkm, "lgkmcnt")):
            for op in q:
                if reg in op["regs"]:
                    raise EmuError("wave %d pc %d: %s%d touched while a load into it is outstanding (%s): %s"
                                   % (w.wid, w.pc, reg[0], reg[1], name, self.cur))
        self._cur_regs.append((reg, read))

    def _check_waitstates(self, w, op, regs_read, regs_written):
        if not self.strict:
            return
        is_mfma = op.startswith("v_mfma")
        for reg in regs_read:
            t = w.wtime.get(reg)
            if not t:
                continue
            dist = w.issue - t[0]
            if t[1] == "mfma" and not is_mfma and dist < MFMA_TO_VALU:
                raise EmuError("wave %d: %s reads %s%d %d slots after the MFMA that writes it: %s" % (w.wid, op, reg[0], reg[1], dist, self.cur))
            if t[1] in ("valu", "trans") and is_mfma and dist <= VALU_TO_MFMA:
                raise EmuError("wave %d: MFMA reads %s%d %d slots after a vector write: %s" % (w.wid, reg[0], reg[1], dist, self.cur))
            if t[1] == "trans" and op.startswith("v_") and dist <= TRANS_TO_VALU:
                raise EmuError("wave %d: %s consumes a transcendental result in the next slot: %s" % (w.wid, op, self.cur))
            if op.startswith("v_permlane") and t[1] in ("valu", "trans") and dist <= VALU_TO_PERM:
                raise EmuError("wave %d: v_permlane reads a just-written register: %s" % (w.wid, self.cur))
        for reg in regs_written:
            t = w.wtime.get(reg)
            if t and t[1] == "mfma" and not is_mfma and w.issue - t[0] < MFMA_TO_VALU:
                raise EmuError("wave %d: %s overwrites %s%d %d slots after an MFMA wrote it: %s" % (w.wid, op, reg[0], reg[1], w.issue - t[0], self.cur))

    # ---- LDS / memory -----------------------------------------------------------------------------------------------------
    def _lds_read(self, w, addr, nbytes):
        out = np.zeros((64, nbytes), dtype=np.uint8)
        lds = self.wg.lds
        for l in range(64):
            a = int(addr[l])
            if a + nbytes > lds.size:
                raise EmuError("wave %d: LDS read out of range (%d): %s" % (w.wid, a, self.cur))
            out[l] = lds[a:a + nbytes]
            if self.strict:
                kb = a >> 10
                st = self.wg.dma_state[kb]
                if st is not None:
                    if st["retired_epoch"] is None:
                        raise EmuError("wave %d: LDS read of KiB %d while wave %d's LDS-DMA into it has not been waited for: %s"
                                       % (w.wid, kb, st["wave"], self.cur))
                    if st["wave"] != w.wid and not (w.epoch > st["retired_epoch"]):
                        raise EmuError("wave %d: LDS read of KiB %d without a barrier after wave %d's vmcnt wait: %s"
                                       % (w.wid, kb, st["wave"], self.cur))
                rd = self.wg.readers[kb].setdefault(w.wid, dict(pending=0, epoch=-1))
                rd["pending"] += 1
                self._cur_kbs.append(kb)
        return out

    def _retire_lds_read(self, w, op):
        """an LDS read of the wave has returned (its s_waitcnt lgkmcnt retired it)"""
        for kb in op.get("kbs", ()):
            rd = self.wg.readers[kb].get(w.wid)
            if rd is not None:
                rd["pending"] -= 1
                rd["epoch"] = w.epoch

    def _dma_write(self, w, lds_base, src_addr):
        lds = self.wg.lds
        if self.strict and w.issue - w.m0_time <= M0_TO_DMA:
            raise EmuError("wave %d: LDS-DMA right behind the write of M0: %s" % (w.wid, self.cur))
        if lds_base % 16:
            raise EmuError("LDS-DMA base not 16-byte aligned")
        kbs = set()
        for l in range(64):
            a = int(src_addr[l])
            d = lds_base + 16 * l
            if a < 0 or a + 16 > self.wg.mem.size or d + 16 > lds.size:
                raise EmuError("wave %d: LDS-DMA out of range (src %d, dst %d): %s" % (w.wid, a, d, self.cur))
            lds[d:d + 16] = self.wg.mem[a:a + 16]
            kbs.add(d >> 10)
        if self.strict:
            for kb in kbs:
                for wid, rd in self.wg.readers[kb].items():
                    if wid == w.wid:
                        if rd["pending"]:
                            raise EmuError("wave %d: LDS-DMA into KiB %d over its own un-waited read: %s" % (w.wid, kb, self.cur))
                    else:
                        # the reader must have finished the read before a barrier this wave has passed since
                        if rd["pending"] or not (w.epoch > rd["epoch"]):
                            raise EmuError("wave %d: LDS-DMA into KiB %d that wave %d read in epoch %d (pending %d) with no barrier since: %s"
                                           % (w.wid, kb, wid, rd["epoch"], rd["pending"], self.cur))
                self.wg.readers[kb] = {}
                self.wg.dma_state[kb] = dict(wave=w.wid, retired_epoch=None, ids=None)
        return kbs

    # ---- execution -------------------------------------------------------------------------------------------------------
    def run(self, max_steps=10_000_000):
        waves = self.wg.waves
        steps = 0
        while True:
            progressed = False
            for w in waves:
                while not w.done and not w.at_barrier:
                    self.step(w)
                    steps += 1
                    progressed = True
                    if steps > max_steps:
                        raise EmuError("step limit")
            if all(w.done for w in waves):
                return
            if all(w.done or w.at_barrier for w in waves):
                if any(w.done for w in waves) and any(w.at_barrier for w in waves):
                    raise EmuError("a wave waits at a barrier other waves never reach")
                for w in waves:
                    w.at_barrier = False
                    w.epoch += 1
                continue
            if not progressed:
                raise EmuError("deadlock")

    def step(self, w):
        prog, labels = self.progs[w.wid]
        if w.pc >= len(prog):
            w.done = True
            return
        op, ops, mods, text = prog[w.pc]
        self.cur = text
        self._cur_regs = []
        self._cur_kbs = []
        w.pc += 1
        w.counts[op] = w.counts.get(op, 0) + 1
        h = getattr(self, "op_" + op, None)
        if h is None:
            raise EmuError("unsupported instruction: " + text)
        h(w, ops, mods)
        if op == "s_nop":
            w.issue += int(ops[0]) + 1
        else:
            w.issue += 1
        if self.strict and op not in ("s_nop", "s_waitcnt", "s_barrier"):
            rd = {r for r, is_read in self._cur_regs if is_read}
            wr = {r for r, is_read in self._cur_regs if not is_read}
            # (the write times were already updated by _vset; check reads against the previous writers kept in _prev)
        # wait-state checks use the register lists recorded by the handlers
        return

    # ---- scalar ------------------------------------------------------------------------------------------------------------
    def op_s_nop(self, w, ops, mods):
        pass

    def op_s_mov_b32(self, w, ops, mods):
        self._sset(w, ops[0], self._sval(w, ops[1]))

    def op_s_movk_i32(self, w, ops, mods):
        self._sset(w, ops[0], self._sval(w, ops[1]))

    def op_s_mov_b64(self, w, ops, mods):
        src = ops[1]
        if parse_reg(src) or src in ("exec", "vcc"):
            val = self._sval(w, src)
        else:
            raise EmuError("s_mov_b64 of a literal is not modelled (extension rules): " + self.cur)
        self._sset(w, ops[0], val)

    def _salu2(self, w, ops, fn, carry=False):
        a, b = self._sval(w, ops[1]), self._sval(w, ops[2])
        r = fn(a, b)
        self._sset(w, ops[0], r)
        return r

    def op_s_add_u32(self, w, ops, mods):
        r = self._sval(w, ops[1]) + self._sval(w, ops[2])
        w.scc = 1 if r > 0xFFFFFFFF else 0
        self._sset(w, ops[0], r)

    def op_s_addc_u32(self, w, ops, mods):
        r = self._sval(w, ops[1]) + self._sval(w, ops[2]) + w.scc
        w.scc = 1 if r > 0xFFFFFFFF else 0
        self._sset(w, ops[0], r)

    def op_s_sub_u32(self, w, ops, mods):
        a, b = self._sval(w, ops[1]), self._sval(w, ops[2])
        w.scc = 1 if b > a else 0
        self._sset(w, ops[0], (a - b) & 0xFFFFFFFF)

    def op_s_mul_i32(self, w, ops, mods):
        self._sset(w, ops[0], (self._sval(w, ops[1]) * self._sval(w, ops[2])) & 0xFFFFFFFF)

    def op_s_min_u32(self, w, ops, mods):
        a, b = self._sval(w, ops[1]), self._sval(w, ops[2])
        w.scc = 1 if a <= b else 0
        self._sset(w, ops[0], min(a, b))

    def op_s_lshl_b32(self, w, ops, mods):
        self._sset(w, ops[0], (self._sval(w, ops[1]) << (self._sval(w, ops[2]) & 31)) & 0xFFFFFFFF)

    def op_s_lshr_b32(self, w, ops, mods):
        self._sset(w, ops[0], self._sval(w, ops[1]) >> (self._sval(w, ops[2]) & 31))

    def op_s_and_b32(self, w, ops, mods):
        self._sset(w, ops[0], self._sval(w, ops[1]) & self._sval(w, ops[2]))

    def op_s_or_b32(self, w, ops, mods):
        self._sset(w, ops[0], self._sval(w, ops[1]) | self._sval(w, ops[2]))

    def op_s_cmp_lt_u32(self, w, ops, mods):
        w.scc = 1 if self._sval(w, ops[0]) < self._sval(w, ops[1]) else 0

    def op_s_cmp_eq_u32(self, w, ops, mods):
        w.scc = 1 if self._sval(w, ops[0]) == self._sval(w, ops[1]) else 0

    def op_s_cmp_lg_u32(self, w, ops, mods):
        w.scc = 1 if self._sval(w, ops[0]) != self._sval(w, ops[1]) else 0

    def op_s_cselect_b32(self, w, ops, mods):
        self._sset(w, ops[0], self._sval(w, ops[1]) if w.scc else self._sval(w, ops[2]))

    def _jump(self, w, label):
        labels = self.progs[w.wid][1]
        if label not in labels:
            raise EmuError("unknown label " + label)
        w.pc = labels[label]

    def op_s_branch(self, w, ops, mods):
        self._jump(w, ops[0])

    def op_s_cbranch_scc1(self, w, ops, mods):
        if w.scc:
            self._jump(w, ops[0])

    def op_s_cbranch_scc0(self, w, ops, mods):
        if not w.scc:
            self._jump(w, ops[0])

    def op_s_cbranch_vccnz(self, w, ops, mods):
        if w.vcc != 0:
            self._jump(w, ops[0])

    def op_s_cbranch_vccz(self, w, ops, mods):
        if w.vcc == 0:
            self._jump(w, ops[0])

    def op_s_barrier(self, w, ops, mods):
        w.at_barrier = True

    def op_s_waitcnt(self, w, ops, mods):
        for o in ops:
            m = re.match(r"(vmcnt|lgkmcnt)\((\d+)\)", o)
            if not m:
                raise EmuError("s_waitcnt operand: " + o)
            n = int(m.group(2))
            if m.group(1) == "vmcnt":
                retire = w.vm[:max(0, len(w.vm) - n)]
                w.vm = w.vm[len(retire):]
                for op in retire:
                    for kb in op.get("kbs", ()):
                        st = self.wg.dma_state[kb]
                        if st is not None and st["wave"] == w.wid and st["retired_epoch"] is None and st["ids"] is op:
                            st["retired_epoch"] = w.epoch
            else:
                if n > 15:
                    raise EmuError("lgkmcnt above 15")
                keep = w.lgkm[max(0, len(w.lgkm) - n):] if n > 0 else []
                for op in w.lgkm[:len(w.lgkm) - len(keep)]:
                    self._retire_lds_read(w, op)
                w.lgkm = keep

    # ---- vector ALU ---------------------------------------------------------------------------------------------------------
    def _regs(self, toks):
        out = []
        for t in toks:
            r = parse_reg(t)
            if r and r[0] in "va":
                out += [(r[0], r[1] + k) for k in range(r[2])]
        return out

    def _valu(self, w, op, dst, srcs, fn, cls="valu", flt=False):
        if self.strict:
            self._check_waitstates(w, op, self._regs(srcs), self._regs([dst]))
        vals = [self._src(w, s, flt) for s in srcs]
        self._vset(w, dst, fn(*vals), cls)

    def op_v_mov_b32(self, w, ops, mods):
        self._valu(w, "v_mov_b32", ops[0], [ops[1]], lambda a: a)

    def op_v_accvgpr_write_b32(self, w, ops, mods):
        self._valu(w, "v_accvgpr_write_b32", ops[0], [ops[1]], lambda a: a)

    def op_v_accvgpr_read_b32(self, w, ops, mods):
        self._valu(w, "v_accvgpr_read_b32", ops[0], [ops[1]], lambda a: a)

    def _fop(self, w, name, ops, fn, cls="valu"):
        with np.errstate(all="ignore"):
            self._valu(w, name, ops[0], ops[1:], lambda *x: u32(fn(*[self._fsrc(t, v) for t, v in zip(ops[1:], x)])), cls, flt=True)

    @staticmethod
    def _fsrc(tok, val):
        # integer inline constants in a float instruction: only 0 is used by the generators (+0.0)
        return f32(val)

    def op_v_add_f32(self, w, ops, mods):
        self._fop(w, "v_add_f32", ops, lambda a, b: a + b)

    def op_v_sub_f32(self, w, ops, mods):
        self._fop(w, "v_sub_f32", ops, lambda a, b: a - b)

    def op_v_mul_f32(self, w, ops, mods):
        self._fop(w, "v_mul_f32", ops, lambda a, b: a * b)

    def op_v_pk_add_f32(self, w, ops, mods):
        """packed fp32: d[0:1] = a[0:1] + b[0:1] (register pairs, default op_sel: lo + lo, hi + hi)"""
        regs = [parse_reg(t) for t in ops[:3]]
        for r, t in zip(regs, ops[:3]):
            if not r or r[0] != "v" or r[2] != 2 or r[1] % 2:
                raise EmuError("v_pk_add_f32: even-aligned v register pairs expected: " + self.cur)
        if self.strict:
            self._check_waitstates(w, "v_pk_add_f32", [("v", r[1] + k) for r in regs[1:] for k in range(2)],
                                   [("v", regs[0][1] + k) for k in range(2)])
        with np.errstate(all="ignore"):
            res = [u32(f32(self._src(w, "v%d" % (regs[1][1] + k))) + f32(self._src(w, "v%d" % (regs[2][1] + k)))) for k in range(2)]
        for k in range(2):
            self._vset(w, "v%d" % (regs[0][1] + k), res[k])

    def op_v_xor_b32_sdwa(self, w, ops, mods):
        """dst = sel0(src0) ^ sel1(src1), whole-dword destination (the generators use src1_sel:WORD_1 for x ^ (x >> 16))"""
        if mods.get("dst_sel", "DWORD") != "DWORD":
            raise EmuError("v_xor_b32_sdwa: partial destination not modelled: " + self.cur)
        if self.strict:
            self._check_waitstates(w, "v_xor_b32_sdwa", self._regs(ops[1:3]), self._regs([ops[0]]))
        vals = []
        for k, t in enumerate(ops[1:3]):
            x = self._src(w, t).astype(np.uint64)
            sel = mods.get("src%d_sel" % k, "DWORD")
            if sel.startswith("BYTE_"):
                x = (x >> (8 * int(sel[5:]))) & 0xFF
            elif sel.startswith("WORD_"):
                x = (x >> (16 * int(sel[5:]))) & 0xFFFF
            vals.append(x)
        self._vset(w, ops[0], (vals[0] ^ vals[1]).astype(np.uint32))

    def op_v_max_f32(self, w, ops, mods):
        self._fop(w, "v_max_f32", ops, lambda a, b: np.fmax(a, b))

    def op_v_max3_f32(self, w, ops, mods):
        self._fop(w, "v_max3_f32", ops, lambda a, b, c: np.fmax(np.fmax(a, b), c))

    def op_v_exp_f32(self, w, ops, mods):
        self._fop(w, "v_exp_f32", ops, lambda a: np.exp2(a.astype(np.float64)).astype(np.float32), "trans")

    def op_v_log_f32(self, w, ops, mods):
        self._fop(w, "v_log_f32", ops, lambda a: np.log2(a.astype(np.float64)).astype(np.float32), "trans")

    def op_v_rcp_f32(self, w, ops, mods):
        self._fop(w, "v_rcp_f32", ops, lambda a: (1.0 / a.astype(np.float64)).astype(np.float32), "trans")

    def _iop(self, w, name, ops, fn):
        self._valu(w, name, ops[0], ops[1:], lambda *x: fn(*[v.astype(np.uint64) for v in x]).astype(np.uint64) & 0xFFFFFFFF)

    def op_v_xor_b32(self, w, ops, mods):
        self._iop(w, "v_xor_b32", ops, lambda a, b: a ^ b)

    def op_v_and_b32(self, w, ops, mods):
        self._iop(w, "v_and_b32", ops, lambda a, b: a & b)

    def op_v_or_b32(self, w, ops, mods):
        self._iop(w, "v_or_b32", ops, lambda a, b: a | b)

    def op_v_add_u32(self, w, ops, mods):
        self._iop(w, "v_add_u32", ops, lambda a, b: a + b)

    def op_v_sub_u32(self, w, ops, mods):
        self._iop(w, "v_sub_u32", ops, lambda a, b: a - b)

    def op_v_mul_lo_u32(self, w, ops, mods):
        self._iop(w, "v_mul_lo_u32", ops, lambda a, b: a * b)

    def op_v_lshrrev_b32(self, w, ops, mods):
        self._iop(w, "v_lshrrev_b32", ops, lambda a, b: b >> (a & 31))

    def op_v_lshlrev_b32(self, w, ops, mods):
        self._iop(w, "v_lshlrev_b32", ops, lambda a, b: b << (a & 31))

    def op_v_lshl_add_u32(self, w, ops, mods):
        self._iop(w, "v_lshl_add_u32", ops, lambda a, b, c: (a << (b & 31)) + c)

    def op_v_bfe_u32(self, w, ops, mods):
        self._iop(w, "v_bfe_u32", ops, lambda a, b, c: (a >> (b & 31)) & ((np.uint64(1) << (c & 31)) - 1))

    def op_v_mbcnt_lo_u32_b32(self, w, ops, mods):
        mask = self._sval(w, ops[1]) if parse_reg(ops[1]) else (0xFFFFFFFF if ops[1] == "-1" else self._imm(ops[1]))
        lanes = np.arange(64)
        cnt = np.array([bin(mask & ((1 << min(l, 32)) - 1)).count("1") for l in lanes], dtype=np.uint64)
        self._valu(w, "v_mbcnt_lo", ops[0], [ops[2]], lambda c: (cnt + c.astype(np.uint64)) & 0xFFFFFFFF)

    def op_v_mbcnt_hi_u32_b32(self, w, ops, mods):
        mask = 0xFFFFFFFF if ops[1] == "-1" else self._imm(ops[1])
        lanes = np.arange(64)
        cnt = np.array([bin(mask & ((1 << max(0, l - 32)) - 1)).count("1") for l in lanes], dtype=np.uint64)
        self._valu(w, "v_mbcnt_hi", ops[0], [ops[2]], lambda c: (cnt + c.astype(np.uint64)) & 0xFFFFFFFF)

    def op_v_cvt_pk_bf16_f32(self, w, ops, mods):
        self._valu(w, "v_cvt_pk_bf16_f32", ops[0], ops[1:], lambda a, b: bf16_round(f32(a)) | (bf16_round(f32(b)) << 16))

    def op_v_cndmask_b32(self, w, ops, mods):
        if ops[3] != "vcc":
            raise EmuError("v_cndmask: vcc expected")
        sel = np.array([(w.vcc >> l) & 1 for l in range(64)], dtype=bool)
        self._valu(w, "v_cndmask_b32", ops[0], ops[1:3], lambda a, b: np.where(sel, b, a))

    def _set_vcc(self, w, cond):
        m = self._mask(w)
        val = 0
        for l in range(64):
            if m[l] and cond[l]:
                val |= 1 << l
        w.vcc = val

    def op_v_cmp_ge_u32_sdwa(self, w, ops, mods):
        if ops[0] != "vcc":
            raise EmuError("sdwa compare: vcc destination expected")
        if self.strict:
            self._check_waitstates(w, "v_cmp", self._regs(ops[1:3]), [])
        vals = []
        for k, t in enumerate(ops[1:3]):
            x = self._src(w, t).astype(np.uint64)
            sel = mods.get("src%d_sel" % k, "DWORD")
            if sel.startswith("BYTE_"):
                x = (x >> (8 * int(sel[5:]))) & 0xFF
            elif sel.startswith("WORD_"):
                x = (x >> (16 * int(sel[5:]))) & 0xFFFF
            vals.append(x)
        self._set_vcc(w, vals[0] >= vals[1])

    def op_v_cmp_nge_f32(self, w, ops, mods):
        if self.strict:
            self._check_waitstates(w, "v_cmp", self._regs(ops[1:3]), [])
        a, b = f32(self._src(w, ops[1], True)), f32(self._src(w, ops[2], True))
        with np.errstate(all="ignore"):
            self._set_vcc(w, ~(a >= b))

    def op_v_permlane32_swap_b32(self, w, ops, mods):
        if self.strict:
            self._check_waitstates(w, "v_permlane32_swap_b32", self._regs(ops[0:2]), self._regs(ops[0:2]))
        d, s = self._src(w, ops[0]), self._src(w, ops[1])
        nd, ns = d.copy(), s.copy()
        nd[32:] = s[:32]
        ns[:32] = d[32:]
        self._vset(w, ops[0], nd)
        self._vset(w, ops[1], ns)

    def op_v_and_b32_dpp(self, w, ops, mods):
        """dst = src0 (lanes permuted within each quad by quad_perm) & src1"""
        qp = [int(c) for c in mods["quad_perm"]]
        lanes = np.arange(64)
        sel = (lanes & ~3) + np.array(qp)[lanes & 3]
        if self.strict:
            self._check_waitstates(w, "v_and_b32_dpp", self._regs(ops[1:3]), self._regs([ops[0]]))
            t = w.wtime.get(parse_reg(ops[1])[:2])
            if t and t[1] in ("valu", "trans") and w.issue - t[0] <= 2:
                raise EmuError("wave %d: DPP reads a register written %d slots ago (needs 2 wait states): %s" % (w.wid, w.issue - t[0], self.cur))
        a, b = self._src(w, ops[1]), self._src(w, ops[2])
        self._vset(w, ops[0], a[sel] & b)

    def op_v_cmp_ge_u32(self, w, ops, mods):
        if self.strict:
            self._check_waitstates(w, "v_cmp", self._regs(ops[1:3]), [])
        a, b = self._src(w, ops[1]).astype(np.uint64), self._src(w, ops[2]).astype(np.uint64)
        self._set_vcc(w, a >= b)

    def op_s_cmp_ge_u32(self, w, ops, mods):
        w.scc = 1 if self._sval(w, ops[0]) >= self._sval(w, ops[1]) else 0

    def op_ds_read_b32(self, w, ops, mods):
        addr = self._lds_addr(w, ops[1], mods)
        if np.any(addr % 4):
            raise EmuError("ds_read_b32 misaligned: " + self.cur)
        kind, idx, n = parse_reg(ops[0])
        self._touch(w, (kind, idx), read=False)
        data = self._lds_read(w, addr, 4).view(np.uint32)
        (w.v if kind == "v" else w.a)[idx] = data[:, 0]
        w.wtime[(kind, idx)] = (w.issue, "lds")
        w.lgkm.append(dict(regs={(kind, idx)}, kbs=list(self._cur_kbs)))

    def op_ds_write_b32(self, w, ops, mods):
        addr = self._lds_addr(w, ops[0], mods)
        kind, idx, n = parse_reg(ops[1])
        if self.strict:
            self._check_waitstates(w, "ds_write_b32", [(kind, idx)], [])
        val = self._src(w, ops[1])
        m = self._mask(w)
        for l in range(64):
            if m[l]:
                a = int(addr[l])
                if a % 4 or a + 4 > self.wg.lds.size:
                    raise EmuError("ds_write_b32 address: " + self.cur)
                self.wg.lds[a:a + 4] = np.frombuffer(struct.pack("<I", int(val[l])), dtype=np.uint8)
        w.lgkm.append(dict(regs=set()))

    def op_buffer_load_dword(self, w, ops, mods):
        """the `offen lds` form: lane l's dword lands at M0 + 4 l"""
        if "lds" not in mods or "offen" not in mods:
            raise EmuError("only the `offen lds` form is modelled: " + self.cur)
        voff = self._src(w, ops[0]).astype(np.int64)
        kind, idx, n = parse_reg(ops[1])
        base = (int(w.s[idx]) & 0xFFFFFFFF) | ((int(w.s[idx + 1]) & 0xFFFF) << 32)
        src = base + voff + self._sval(w, ops[2]) + int(mods.get("offset", 0))
        if self.strict and w.issue - w.m0_time <= M0_TO_DMA:
            raise EmuError("wave %d: LDS-DMA right behind the write of M0: %s" % (w.wid, self.cur))
        kbs = set()
        for l in range(64):
            a, d = int(src[l]), w.m0 + 4 * l
            if a < 0 or a + 4 > self.wg.mem.size or d + 4 > self.wg.lds.size:
                raise EmuError("wave %d: LDS-DMA (dword) out of range: %s" % (w.wid, self.cur))
            self.wg.lds[d:d + 4] = self.wg.mem[a:a + 4]
            kbs.add(d >> 10)
        op = dict(regs=set(), kbs=kbs)
        if self.strict:
            for kb in kbs:
                for wid, rd in self.wg.readers[kb].items():
                    if rd["pending"] and wid == w.wid:
                        raise EmuError("wave %d: LDS-DMA into KiB %d over its own un-waited read: %s" % (w.wid, kb, self.cur))
                    if wid != w.wid and (rd["pending"] or not (w.epoch > rd["epoch"])):
                        raise EmuError("wave %d: LDS-DMA into KiB %d that wave %d read with no barrier since: %s" % (w.wid, kb, wid, self.cur))
                self.wg.readers[kb] = {}
                self.wg.dma_state[kb] = dict(wave=w.wid, retired_epoch=None, ids=op)
        w.vm.append(op)

    # ---- matrix ------------------------------------------------------------------------------------------------------------
    def op_v_mfma_f32_32x32x16_bf16(self, w, ops, mods):
        dk, di, dn = parse_reg(ops[0])
        ak, ai, an = parse_reg(ops[1])
        bk, bi, bn = parse_reg(ops[2])
        c = parse_reg(ops[3])
        if dn != 16 or an != 4 or bn != 4:
            raise EmuError("mfma operand widths: " + self.cur)
        if w.exec != (1 << 64) - 1:
            raise EmuError("mfma with partial exec")
        rd = [(ak, ai + k) for k in range(4)] + [(bk, bi + k) for k in range(4)]
        if c:
            if c[0] != dk:
                raise EmuError("mfma: C and D must be in the same register file: " + self.cur)
            if c[2] != 16:
                raise EmuError("mfma C width")
            if (c[0], c[1]) != (dk, di):
                rd += [(c[0], c[1] + k) for k in range(16)]      # same register as D: accumulate chain, no software wait
        if self.strict:
            self._check_waitstates(w, "v_mfma", rd, [])
        for r in rd + ([(dk, di + k) for k in range(16)]):
            self._touch(w, r, read=True)
        fa = (w.v if ak == "v" else w.a)[ai:ai + 4]               # [4][64]
        fb = (w.v if bk == "v" else w.a)[bi:bi + 4]
        lanes = np.arange(64)
        r_, h_ = lanes & 31, lanes >> 5
        Am = np.zeros((32, 16), dtype=np.float32)
        Bm = np.zeros((16, 32), dtype=np.float32)
        for j in range(8):
            av = bf16_to_f32(fa[j // 2] >> (16 * (j % 2)))
            bv = bf16_to_f32(fb[j // 2] >> (16 * (j % 2)))
            Am[r_, 8 * h_ + j] = av
            Bm[8 * h_ + j, r_] = bv
        D = Am.astype(np.float64) @ Bm.astype(np.float64)
        file = w.v if dk == "v" else w.a
        for i in range(16):
            rows = (i & 3) + 8 * (i >> 2) + 4 * h_
            cin = f32((w.v if c[0] == "v" else w.a)[c[1] + i]) if c else np.float32(f32(np.uint32(self._imm(ops[3], True))))
            with np.errstate(all="ignore"):
                file[di + i] = u32((D[rows, r_] + cin.astype(np.float64)).astype(np.float32))
            w.wtime[(dk, di + i)] = (w.issue, "mfma")

    # ---- LDS ---------------------------------------------------------------------------------------------------------------
    def _lds_addr(self, w, tok, mods):
        return self._src(w, tok).astype(np.int64) + int(mods.get("offset", 0))

    def _load_pending(self, w, queue, dst):
        kind, idx, n = parse_reg(dst)
        regs = {(kind, idx + k) for k in range(n)}
        queue.append(dict(regs=regs))
        # values are written now; the pending set makes any touch before the wait an error
        return kind, idx, n

    def op_ds_read_b128(self, w, ops, mods):
        addr = self._lds_addr(w, ops[1], mods)
        if np.any(addr % 16):
            raise EmuError("ds_read_b128 misaligned: " + self.cur)
        kind, idx, n = parse_reg(ops[0])
        for k in range(4):
            self._touch(w, (kind, idx + k), read=False)
        data = self._lds_read(w, addr, 16).view(np.uint32)        # [64][4]
        file = w.v if kind == "v" else w.a
        for k in range(4):
            file[idx + k] = data[:, k]
            w.wtime[(kind, idx + k)] = (w.issue, "lds")
        w.lgkm.append(dict(regs={(kind, idx + k) for k in range(4)}, kbs=list(self._cur_kbs)))

    def op_ds_read_b64_tr_b16(self, w, ops, mods):
        if w.exec != (1 << 64) - 1:
            raise EmuError("ds_read_b64_tr_b16 with partial exec")
        addr = self._lds_addr(w, ops[1], mods)
        if np.any(addr % 8):
            raise EmuError("ds_read_b64_tr_b16 misaligned: " + self.cur)
        kind, idx, n = parse_reg(ops[0])
        for k in range(2):
            self._touch(w, (kind, idx + k), read=False)
        raw = self._lds_read(w, addr, 8).view(np.uint16)          # [64][4]: what each lane's address holds
        out = np.zeros((64, 4), dtype=np.uint16)
        for l in range(64):
            g, i = l >> 4, l & 15
            for q in range(4):
                out[l, q] = raw[16 * g + 4 * q + (i >> 2), i & 3]
        file = w.v if kind == "v" else w.a
        file[idx] = out[:, 0].astype(np.uint32) | (out[:, 1].astype(np.uint32) << 16)
        file[idx + 1] = out[:, 2].astype(np.uint32) | (out[:, 3].astype(np.uint32) << 16)
        for k in range(2):
            w.wtime[(kind, idx + k)] = (w.issue, "lds")
        w.lgkm.append(dict(regs={(kind, idx + k) for k in range(2)}, kbs=list(self._cur_kbs)))

    def op_ds_write_b64(self, w, ops, mods):
        addr = self._lds_addr(w, ops[0], mods)
        kind, idx, n = parse_reg(ops[1])
        if self.strict:
            self._check_waitstates(w, "ds_write_b64", [(kind, idx), (kind, idx + 1)], [])
        lo, hi = self._src(w, "%s%d" % (kind, idx)), self._src(w, "%s%d" % (kind, idx + 1))
        m = self._mask(w)
        for l in range(64):
            if m[l]:
                a = int(addr[l])
                if a % 8:
                    raise EmuError("ds_write_b64 misaligned")
                self.wg.lds[a:a + 4] = np.frombuffer(struct.pack("<I", int(lo[l])), dtype=np.uint8)
                self.wg.lds[a + 4:a + 8] = np.frombuffer(struct.pack("<I", int(hi[l])), dtype=np.uint8)
                if self.strict:
                    st = self.wg.dma_state[a >> 10]
                    if st is not None and st["retired_epoch"] is None:
                        raise EmuError("ds_write into a KiB with an LDS-DMA in flight: " + self.cur)
        w.lgkm.append(dict(regs=set()))

    # ---- global memory -----------------------------------------------------------------------------------------------------
    def op_buffer_load_dwordx4(self, w, ops, mods):
        if "lds" not in mods or "offen" not in mods:
            raise EmuError("only the `offen lds` form is modelled: " + self.cur)
        voff = self._src(w, ops[0]).astype(np.int64)
        kind, idx, n = parse_reg(ops[1])
        base = (int(w.s[idx]) & 0xFFFFFFFF) | ((int(w.s[idx + 1]) & 0xFFFF) << 32)
        nrec = int(w.s[idx + 2]) & 0xFFFFFFFF
        soff = self._sval(w, ops[2])
        if np.any(voff + int(mods.get("offset", 0)) + 16 > nrec):
            raise EmuError("buffer load out of the descriptor's range: " + self.cur)
        src = base + voff + soff + int(mods.get("offset", 0))
        kbs = self._dma_write(w, w.m0 & 0xFFFF if False else w.m0, src)
        op = dict(regs=set(), kbs=kbs)
        if self.strict:
            for kb in kbs:
                self.wg.dma_state[kb]["ids"] = op
        w.vm.append(op)

    def op_global_load_dwordx4(self, w, ops, mods):
        kind, idx, n = parse_reg(ops[0])
        voff = self._src(w, ops[1]).astype(np.int64)
        base = self._sval(w, ops[2])
        src = base + voff + int(mods.get("offset", 0))
        file = w.v if kind == "v" else w.a
        for k in range(4):
            self._touch(w, (kind, idx + k), read=False)
        for l in range(64):
            a = int(src[l])
            if a < 0 or a + 16 > self.wg.mem.size:
                raise EmuError("global load out of range: " + self.cur)
            d = self.wg.mem[a:a + 16].view(np.uint32)
            for k in range(4):
                file[idx + k][l] = d[k]
        w.vm.append(dict(regs={(kind, idx + k) for k in range(4)}))

    def op_global_load_dword(self, w, ops, mods):
        kind, idx, n = parse_reg(ops[0])
        voff = self._src(w, ops[1]).astype(np.int64)
        src = self._sval(w, ops[2]) + voff + int(mods.get("offset", 0))
        file = w.v if kind == "v" else w.a
        self._touch(w, (kind, idx), read=False)
        for l in range(64):
            a = int(src[l])
            if a < 0 or a + 4 > self.wg.mem.size:
                raise EmuError("global load out of range: " + self.cur)
            file[idx][l] = self.wg.mem[a:a + 4].view(np.uint32)[0]
        w.vm.append(dict(regs={(kind, idx)}))

    def _gstore(self, w, ops, mods, ndw):
        voff = self._src(w, ops[0]).astype(np.int64)
        kind, idx, n = parse_reg(ops[1])
        if self.strict:
            self._check_waitstates(w, "global_store", [(kind, idx + k) for k in range(ndw)], [])
        base = self._sval(w, ops[2])
        dst = base + voff + int(mods.get("offset", 0))
        m = self._mask(w)
        vals = [self._src(w, "%s%d" % (kind, idx + k)) for k in range(ndw)]
        for l in range(64):
            if m[l]:
                a = int(dst[l])
                if a < 0 or a + 4 * ndw > self.wg.mem.size:
                    raise EmuError("global store out of range: " + self.cur)
                for k in range(ndw):
                    self.wg.mem[a + 4 * k:a + 4 * k + 4] = np.frombuffer(struct.pack("<I", int(vals[k][l])), dtype=np.uint8)
        w.vm.append(dict(regs=set()))

    def op_global_store_dwordx4(self, w, ops, mods):
        self._gstore(w, ops, mods, 4)

    def op_global_store_dword(self, w, ops, mods):
        self._gstore(w, ops, mods, 1)
