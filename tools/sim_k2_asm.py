#!/usr/bin/env python3
"""CPU model of csrc/k2_loop_p12.inc (the hand-scheduled 12-plane stage loop of k_mh_compare_a12).

Interprets the generated instruction stream for ONE lane (scalar control flow, LDS reads, v_xor / v_bitop3 /
v_bcnt / v_lshl_add, the counter write-back); the DMA is modelled as "the stage's 12 KiB image appears in the ring
slot when its first piece is issued", waits and barriers are ignored.  It checks what a timing-free model can check:
the register map, the LDS addressing of both operands, the ring-slot bookkeeping over 16 stages, the popcount
packing and the write-back order -- against a direct evaluation of  sum_g popcount(OR_p (a_p xor b_p)).
Used by tests/test_k2_asm_model.py (no GPU needed)."""
import os
import random
import re
import sys

INC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynaalign_amd", "csrc", "k2_loop_p12.inc")


INC_P = os.path.join(os.path.dirname(INC), "k2_loop_p12p.inc")


INC_16 = os.path.join(os.path.dirname(INC), "k2_loop_p16.inc")


def run16(tx=5, ty=11, seed=1, ns=16):
    """The 16-plane block (K2ASM_PLANES=16, k_mh_compare_a16): padded 80-byte slots, ring of two stages, five DMA pieces"""
    return run_persistent(tx, ty, seed, ntiles=1, ns=ns, inc=INC_16, planes=16, slot_bytes=80, ring=2)


INC_14 = os.path.join(os.path.dirname(INC), "k2_loop_p14.inc")


def run14(tx=5, ty=11, seed=1, ns=16):
    """The seven-step form of the 16-plane block (K2ASM_PLANES=14): same padded slots and ring, plane words 14 and 15 never read"""
    return run_persistent(tx, ty, seed, ntiles=1, ns=ns, inc=INC_14, planes=14, slot_bytes=80, ring=2)


INC_15 = os.path.join(os.path.dirname(INC), "k2_loop_p15.inc")


def run15(tx=5, ty=11, seed=1, ns=16):
    """15 code bits (K2ASM_PLANES=15): eight steps, the last one on plane 14 only -- plane word 15 is read but never used"""
    return run_persistent(tx, ty, seed, ntiles=1, ns=ns, inc=INC_15, planes=15, slot_bytes=80, ring=2)


INC_Q = os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments", "k2_loop_p12q.inc")   # (experiment, not in the product build)


def run_inloop(tx=5, ty=11, seed=1, ntiles=3, ns=16, n_hash=None):
    """The block of k_mh_compare_q12 (K2ASM_INLOOP=1): like run_persistent, plus (1) the previous tile's 64 float64 stores issued from
    inside the stage loop are checked -- address and data of every one, against the reference counts of the previous tile --
    and (2) the counted vmcnt waits are checked against an in-order model of the outstanding vector-memory operations: at every
    stage barrier the three DMA pieces of the stage about to be read must have retired.  Returns (stages issued, wrong counters,
    wrong or missing stores)."""
    return run_persistent(tx, ty, seed, ntiles, ns, inc=INC_Q, inloop=True, n_hash=n_hash or 32 * ns)


def run_persistent(tx=5, ty=11, seed=1, ntiles=3, ns=16, inc=INC_P, planes=12, slot_bytes=48, ring=3, inloop=False, n_hash=500):
    """The block of the persistent kernel (K2ASM_PERSIST=1), executed `ntiles` times in a row on one LDS image, the
    way k_mh_compare_p12 calls it: flags first / has-next, ring phase advancing by ns stages per tile, the next tile's
    source handed in.  The DMA is modelled by ADDRESS: the stage image that appears in the ring is the one the running
    source base points at, so a wrong base switch or ring slot shows up as wrong counters.  Returns (stages issued,
    wrong counters)."""
    lines = [l.strip()[1:].split('\\n')[0] for l in open(inc) if l.startswith('"')]
    STAGE = 256 * slot_bytes; LB = 4096; ST = 128 * slot_bytes; TILE_STRIDE = 1 << 24
    pieces = 64 * slot_bytes // 1024
    random.seed(seed)
    # (in-loop form: sparse plane words, so that the match counts -- and with them the stored values -- vary)
    rb = (lambda: random.getrandbits(8) if random.random() < 0.02 else 0) if inloop else (lambda: random.getrandbits(8))
    stages = [[bytes(rb() for _ in range(STAGE)) for _ in range(ns)] for _ in range(ntiles)]
    TB = LB + ring * STAGE                        # (in-loop form) the count -> double table: entry c = the 8 bytes (c, ~c)
    lds = bytearray(TB + 8 * 1024)
    for c in range(1024):
        lds[TB + 8 * c:TB + 8 * c + 8] = c.to_bytes(4, 'little') + (c ^ 0xffffffff).to_bytes(4, 'little')
    LD = 100096; OUT = 1 << 40                    # (in-loop form) leading dimension and base address of the result matrix
    prev_counts = None; prev_tile = None; bad_stores = 0; pw = 0
    Vkeep = [0] * 256
    vm = []                                       # outstanding vector-memory operations, oldest first: ('dma', tile, stage, piece) / ('st',)
    retired = set()                               # ... carried from block to block: a tile's first two stages are issued by the block before
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\d+):$', l)
        if m: labels.setdefault(m.group(1), []).append(i)
    issued_total = 0
    bad = 0
    phase = 0
    for t in range(ntiles):
        V = [0] * 256; S = {}
        S['lb'] = LB; S['ns'] = ns; S['st'] = ST; S['wv'] = 0
        S['fl'] = (1 if t == 0 else 0) | (2 if t + 1 < ntiles else 0)
        stores = []; gpr_idx = None
        if inloop:
            for r_ in range(64, 96): V[r_] = Vkeep[r_]          # the previous tile's counters come in in v64..v95
            I0, J0 = (3 + 2 * t) * 128, (40 + 3 * t) * 128      # this tile's position (any interior off-diagonal one)
            if prev_tile is not None: S['fl'] |= 4
            pi, pj = prev_tile if prev_tile is not None else (0, 0)
            od = OUT + (pi * LD + pj) * 8; om = OUT + (pj * LD + pi) * 8
            S.update(tb=TB, nn=n_hash * 0x10001, odl=od & 0xffffffff, odh=od >> 32, oml=om & 0xffffffff, omh=om >> 32, l8=LD * 8, pw=pw,
                     kk=16 // ns)
            V[119] = ((2 * ty) * LD + 2 * tx) * 8; V[127] = ((2 * tx) * LD + 2 * ty) * 8
        S['sp'] = phase * STAGE
        base = (t + 1) * TILE_STRIDE; nbase = (t + 2) * TILE_STRIDE
        S['sl'], S['sh'] = base & 0xffffffff, base >> 32
        S['nl'], S['nh'] = nbase & 0xffffffff, nbase >> 32
        V[120] = LB + ty * slot_bytes; V[121] = LB + (128 + tx) * slot_bytes; V[124] = 0
        def sval(x):
            x = x.strip()
            if x.startswith('%['): return S[x[2:-1]]
            if x == 'm0': return S.get('m0', 0)
            if x.startswith('s'): return S.get(x, 0)
            return int(x, 0)
        def vreg(x): return int(x.strip()[1:])
        def vpair(x): return int(re.match(r'v\[(\d+):(\d+)\]', x.strip()).group(1))
        pc = 0; scc = 0; piece = 0
        while pc < len(lines):
            l = lines[pc]; pc += 1
            if re.match(r'^\d+:$', l) or l.startswith('//') or not l: continue
            op, _, rest = l.partition(' ')
            a = [x.strip() for x in rest.split(',')] if rest else []
            if op == 's_mov_b32': S[a[0]] = sval(a[1])
            elif op == 's_add_u32':
                r = sval(a[1]) + sval(a[2]); S[a[0]] = r & 0xffffffff; scc = int(r > 0xffffffff)
            elif op == 's_addc_u32':
                r = sval(a[1]) + sval(a[2]) + scc; S[a[0]] = r & 0xffffffff; scc = int(r > 0xffffffff)
            elif op == 's_sub_u32': S[a[0]] = (sval(a[1]) - sval(a[2])) & 0xffffffff
            elif op == 's_xor_b32': S[a[0]] = sval(a[1]) ^ sval(a[2])
            elif op == 's_mul_i32': S[a[0]] = (sval(a[1]) * sval(a[2])) & 0xffffffff
            elif op == 's_cmp_lt_u32': scc = int(sval(a[0]) < sval(a[1]))
            elif op == 's_cmp_eq_u32': scc = int(sval(a[0]) == sval(a[1]))
            elif op == 's_bitcmp1_b32': scc = (sval(a[0]) >> sval(a[1])) & 1
            elif op == 's_cselect_b32': S[a[0]] = sval(a[1]) if scc else sval(a[2])
            elif op in ('s_cbranch_scc0', 's_cbranch_scc1', 's_branch'):
                take = (op == 's_branch') or (op == 's_cbranch_scc0' and not scc) or (op == 's_cbranch_scc1' and scc)
                if take:
                    num, d = a[0][:-1], a[0][-1]
                    cands = labels[num]
                    pc = min(c for c in cands if c >= pc) if d == 'f' else max(c for c in cands if c < pc)
            elif op == 's_waitcnt':
                m = re.match(r'vmcnt\((\d+)\)', a[0])
                if m and inloop:
                    keep = int(m.group(1))
                    done, vm = (vm, []) if keep == 0 else (vm[:-keep], vm[-keep:])
                    retired.update(x for x in done if x[0] == 'dma')
            elif op == 's_barrier':
                if inloop:                        # the stage about to be read: all three pieces must have landed
                    for q in range(pieces):
                        assert ('dma', t, S.get('s40', 0), q) in retired, "stage %d of tile %d read before its DMA piece %d retired (vmcnt wait too weak)" % (S.get('s40', 0), t, q)
            elif op in ('s_nop', 's_setprio'): pass
            elif op == 's_lshr_b32': S[a[0]] = sval(a[1]) >> sval(a[2])
            elif op == 's_lshl_b32': S[a[0]] = (sval(a[1]) << sval(a[2])) & 0xffffffff
            elif op == 's_and_b32': S[a[0]] = sval(a[1]) & sval(a[2])
            elif op == 's_set_gpr_idx_on': gpr_idx = sval(a[0]); S['m0'] = 0xdead0000 | gpr_idx   # (m0 is clobbered)
            elif op == 's_set_gpr_idx_off': gpr_idx = None
            elif op == 'v_mov_b32':
                if a[1].startswith('v'): V[vreg(a[0])] = V[vreg(a[1]) + (gpr_idx or 0)]
                else: V[vreg(a[0])] = int(a[1], 0)
            elif op == 'v_and_b32': V[vreg(a[0])] = int(a[1], 0) & V[vreg(a[2])]
            elif op == 'v_lshlrev_b32': V[vreg(a[0])] = (V[vreg(a[2])] << int(a[1])) & 0xffffffff
            elif op == 'v_lshrrev_b32': V[vreg(a[0])] = V[vreg(a[2])] >> int(a[1])
            elif op == 'v_sub_u32': V[vreg(a[0])] = (sval(a[1]) - V[vreg(a[2])]) & 0xffffffff
            elif op == 'v_swap_b32': V[vreg(a[0])], V[vreg(a[1])] = V[vreg(a[1])], V[vreg(a[0])]
            elif op == 'global_store_dwordx4':
                m = re.match(r's\[(\d+):(\d+)\]', a[2].split()[0])
                addr = (S['s' + m.group(1)] | (S['s' + m.group(2)] << 32)) + V[vreg(a[0])]
                b = vpair(a[1])
                stores.append((addr, tuple(V[b:b + 4])))
                vm.append(('st',))
            elif op == 'global_load_lds_dwordx4':
                m = re.match(r's\[(\d+):(\d+)\]', a[1])
                addr = S['s' + m.group(1)] | (S['s' + m.group(2)] << 32)
                if piece == 0:
                    tt, off = addr // TILE_STRIDE - 1, addr % TILE_STRIDE
                    assert off % ST == 0 and 0 <= tt < ntiles and off // ST < ns, "DMA source outside any stage: %x" % addr
                    slot, rem = divmod(S['m0'] - LB, STAGE)
                    assert rem == 0 and 0 <= slot < ring, "DMA destination outside the ring: %x" % S['m0']
                    lds[LB + slot * STAGE:LB + (slot + 1) * STAGE] = stages[tt][off // ST]
                    issued_total += 1
                    dma_tag = (tt, off // ST)
                else:
                    assert addr % TILE_STRIDE % ST == piece * 1024
                vm.append(('dma', dma_tag[0], dma_tag[1], piece))
                piece = (piece + 1) % pieces
            elif op == 'v_add_u32':
                x = sval(a[1]) if not a[1].startswith('v') else V[vreg(a[1])]
                V[vreg(a[0])] = (x + V[vreg(a[2])]) & 0xffffffff
            elif op == 'ds_read_b64':
                b = vpair(a[0]); addr = V[vreg(a[1].split()[0])]; mo = re.search(r'offset:(\d+)', l); off = int(mo.group(1)) if mo else 0
                ad = addr + off
                V[b] = int.from_bytes(lds[ad:ad + 4], 'little'); V[b + 1] = int.from_bytes(lds[ad + 4:ad + 8], 'little')
            elif op == 'v_xor_b32': V[vreg(a[0])] = V[vreg(a[1])] ^ V[vreg(a[2])]
            elif op == 'v_bitop3_b32':
                d_, s0, s1, s2 = vreg(a[0]), vreg(a[1]), vreg(a[2]), vreg(a[3].split()[0])
                V[d_] = V[s0] | (V[s1] ^ V[s2])
            elif op == 'v_bcnt_u32_b32':
                x = V[vreg(a[2])] if a[2].startswith('v') else int(a[2], 0)
                V[vreg(a[0])] = bin(V[vreg(a[1])]).count('1') + x
            elif op == 'v_lshl_add_u32': V[vreg(a[0])] = ((V[vreg(a[1])] << int(a[2])) + V[vreg(a[3])]) & 0xffffffff
            else:
                raise RuntimeError("unhandled instruction: " + l)
        def word(stage, slot, w):
            o = slot * slot_bytes + w * 4; return int.from_bytes(stages[t][stage][o:o + 4], 'little')
        counts_now = {}
        for r in range(8):
            for c in range(8):
                arow = r * 16 + ty; bcol = 128 + c * 16 + tx
                mism = 0
                for st in range(ns):
                    dd = 0
                    for pl in range(planes):
                        dd |= word(st, arow, pl) ^ word(st, bcol, pl ^ 1)
                    mism += bin(dd).count('1')
                if ((V[64 + 4 * r + c // 2] >> (16 * (c & 1))) & 0xffff) != mism: bad += 1
                if inloop: counts_now[(r, c)] = mism
        if inloop:
            # DMA pieces of the NEXT tile's first stages issued here retire in the next block: carry them over as already retired
            # only if that block's waits say so -- modelled by handing the outstanding list on
            want = {}
            if prev_counts is not None:
                def tab(c): return (c, c ^ 0xffffffff)
                pi, pj = prev_tile
                for rp in range(4):
                    for c2 in range(4):
                        m = {(e, f): n_hash - prev_counts[(2 * rp + e, 2 * c2 + f)] for e in (0, 1) for f in (0, 1)}
                        for e in (0, 1):      # direct rows 32 rp + 2 ty + e, columns 32 c2 + 2 tx, + 1
                            want[OUT + ((pi + 32 * rp + 2 * ty + e) * LD + pj + 32 * c2 + 2 * tx) * 8] = tab(m[(e, 0)]) + tab(m[(e, 1)])
                        for f in (0, 1):      # mirrored rows 32 c2 + 2 tx + f, columns 32 rp + 2 ty, + 1
                            want[OUT + ((pj + 32 * c2 + 2 * tx + f) * LD + pi + 32 * rp + 2 * ty) * 8] = tab(m[(0, f)]) + tab(m[(1, f)])
            got = dict(stores)
            bad_stores += sum(1 for k_, v_ in want.items() if got.get(k_) != v_) + sum(1 for k_ in got if k_ not in want) + (len(stores) - len(got))
            prev_counts, prev_tile = counts_now, (I0, J0)
            pw = (4 * (16 // ns)) if (S['fl'] & 4) else 0
            Vkeep = list(V)
        phase = (phase + ns) % ring
    return (issued_total, bad, bad_stores) if inloop else (issued_total, bad)


def run(tx=5, ty=11, seed=1, inc=INC):
    lines=[l.strip()[1:].split('\\n')[0] for l in open(inc) if l.startswith('"')]
    NS=16; STAGE=12288; LB=0; ST=6144
    random.seed(seed)
    # stage data: for each stage s: 256 slots x 48 bytes. a slots 0..127, b slots 128..255
    stages=[bytes(random.getrandbits(8) for _ in range(STAGE)) for s in range(NS)]
    lds=bytearray(3*STAGE+4096)
    V=[0]*256; S={}
    S['lb']=LB; S['ns']=NS; S['st']=ST; S['wv']=0
    V[120]=LB+ty*48; V[121]=LB+(128*3+tx*3)*16; V[122]=0; V[123]=0; V[124]=0  # tid 0 -> wave 0 (only data path matters)
    labels={}
    for i,l in enumerate(lines):
        m=re.match(r'^(\d+):$',l)
        if m: labels.setdefault(m.group(1),[]).append(i)
    def sval(x):
        x=x.strip()
        if x.startswith('%['): return S[x[2:-1]]
        if x.startswith('s'): return S.get(x,0)
        if x=='m0': return S.get('m0',0)
        return int(x,0)
    def vreg(x): return int(x.strip()[1:])
    def vpair(x):
        m=re.match(r'v\[(\d+):(\d+)\]',x.strip()); return int(m.group(1))
    pc=0; scc=0; issued=0; dma_q=0; wb={}
    steps=0
    while pc<len(lines):
        l=lines[pc]; pc+=1
        if re.match(r'^\d+:$',l) or l.startswith('//') or not l: continue
        op,_,rest=l.partition(' ')
        a=[x.strip() for x in rest.split(',')] if rest else []
        if op=='s_mov_b32': S[a[0]]=sval(a[1])
        elif op=='s_add_u32': S[a[0]]=(sval(a[1])+sval(a[2]))&0xffffffff
        elif op=='s_sub_u32': S[a[0]]=(sval(a[1])-sval(a[2]))&0xffffffff
        elif op=='s_mul_i32': S[a[0]]=(sval(a[1])*sval(a[2]))&0xffffffff
        elif op=='s_cmp_lt_u32': scc=int(sval(a[0])<sval(a[1]))
        elif op=='s_cmp_eq_u32': scc=int(sval(a[0])==sval(a[1]))
        elif op=='s_cselect_b32': S[a[0]]=sval(a[1]) if scc else sval(a[2])
        elif op in('s_cbranch_scc0','s_cbranch_scc1','s_branch'):
            take = (op=='s_branch') or (op=='s_cbranch_scc0' and not scc) or (op=='s_cbranch_scc1' and scc)
            if take:
                t=a[0]; num=t[:-1]; d=t[-1]
                cands=labels[num]
                if d=='f': pc=min(c for c in cands if c>=pc)
                else: pc=max(c for c in cands if c<pc)
        elif op in('s_nop','s_waitcnt','s_barrier','s_setprio'): pass
        elif op=='v_mov_b32': V[vreg(a[0])]= V[vreg(a[1])] if a[1].startswith('v') else int(a[1],0)
        elif op=='v_lshrrev_b32': V[vreg(a[0])]=V[vreg(a[2])]>>int(a[1])
        elif op=='v_readfirstlane_b32': S[a[0]]=V[vreg(a[1])]
        elif op in('v_add_co_u32','v_addc_co_u32'): pass
        elif op=='global_load_lds_dwordx4':
            # emulate: whole stage copied when its first piece is issued (single-lane data-path model)
            if dma_q==0:
                s=issued; slot=(S['m0']-S['lb'])//STAGE   # wave 0: m0 = lb + slot*STAGE (+0)
                lds[slot*STAGE:(slot+1)*STAGE]=stages[s]
                issued+=1
            dma_q=(dma_q+1)%3
        elif op=='v_add_u32':
            x=sval(a[1]) if not a[1].startswith('v') else V[vreg(a[1])]
            V[vreg(a[0])]=(x+V[vreg(a[2])])&0xffffffff
        elif op=='ds_read_b64':
            base=vpair(a[0]); addr=V[vreg(a[1].split()[0])]; off=int(re.search(r'offset:(\d+)',l).group(1))
            ad=addr+off-LB
            V[base]=int.from_bytes(lds[ad:ad+4],'little'); V[base+1]=int.from_bytes(lds[ad+4:ad+8],'little')
        elif op=='v_xor_b32': V[vreg(a[0])]=V[vreg(a[1])]^V[vreg(a[2])]
        elif op=='v_bitop3_b32':
            d_,s0,s1,s2=vreg(a[0]),vreg(a[1]),vreg(a[2]),vreg(a[3].split()[0])
            V[d_]=V[s0]|(V[s1]^V[s2])
        elif op=='v_bcnt_u32_b32':
            x=V[vreg(a[2])] if a[2].startswith('v') else int(a[2],0)
            V[vreg(a[0])]=bin(V[vreg(a[1])]).count('1')+x
        elif op=='v_lshl_add_u32': V[vreg(a[0])]=((V[vreg(a[1])]<<int(a[2]))+V[vreg(a[3])])&0xffffffff
        elif op=='ds_write_b32':
            off=int(re.search(r'offset:(\d+)',l).group(1)); wb[off//1024]=V[vreg(a[1].split()[0])]
        else:
            raise RuntimeError("unhandled instruction: " + l)
    # reference
    def word(stage,slot,w): 
        o=slot*48+w*4; return int.from_bytes(stages[stage][o:o+4],'little')
    bad=0
    for r in range(8):
        for c in range(8):
            arow=r*16+ty; bcol=128+c*16+tx
            mism=0
            for s in range(NS):
                dd=0
                for p in range(12):
                    aw=word(s,arow,p); bw=word(s,bcol,p^1)
                    dd|=aw^bw
                mism+=bin(dd).count('1')
            packed = wb[4*r+c//2] if wb else V[64+4*r+c//2]      # through LDS, or left in v64..v95 (K2ASM_REGOUT)
            got=(packed>>(16*(c&1)))&0xffff
            if got!=mism: bad+=1
    return issued, bad



if __name__ == "__main__":
    issued, bad = run()
    print("stages issued", issued, "wrong counters", bad)
    pissued, pbad = run_persistent()
    print("persistent block, 3 tiles: stages issued", pissued, "wrong counters", pbad)
    i16, b16 = run16()
    print("16-plane block: stages issued", i16, "wrong counters", b16)
    sys.exit(1 if bad or issued != 16 or pbad or pissued != 48 or b16 or i16 != 16 else 0)
