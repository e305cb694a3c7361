#!/usr/bin/env python3
"""Analysis (not a test, CPU only): how deep are the chains of copies-of-copies in liblz4's block-linked frames, and what does pointer jumping buy?

For the first blocks of an 8 MiB Silesia-like chunk compressed the way a stock reference sender does (lz4.frame.compress defaults, oracle/ref.py), every match
becomes a descriptor (destination, offset, length); depth[d] = 1 + the maximum depth of the matches whose destinations d's source touches.  That depth -- not
the number of matches -- bounds sky_lz4_link (csrc/lz4d_kernel.inc): a level costs a wavefront ~800 cycles whatever the lane count.  Then the shortening the
kernel applies is simulated: a short plain copy whose source lies inside the destination of ONE earlier plain copy reads that copy's source instead and
inherits its range.  Printed per block: matches, then (maximum depth, mean depth, redirections) after 0 / 1 / 2 / 4 / 12 rounds.
Numbers of round 3: profiles/r3_linked_decode.txt."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, struct, bisect
from oracle import ref
from skyplane_amd import synth
cb = synth.CHUNK_BYTES
unit = synth.silesia_like(2 * cb, config_id=2)
data = unit[:cb].tobytes()
f = ref.lz4f_compress(data)
flg = f[4]; pos = 6 + (8 if flg & 8 else 0) + 1
res = []
nblk = 0
while True:
    bs, = struct.unpack_from('<I', f, pos); pos += 4
    if bs == 0: break
    raw = bs >> 31; sz = bs & 0x7fffffff
    blk = f[pos:pos+sz]; pos += sz
    nblk += 1
    if raw or nblk > 24: continue
    ip = 0; op = 0; n = len(blk)
    D = []   # (mop, off, ml)
    while ip < n:
        tok = blk[ip]; ip += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = blk[ip]; ip += 1; lit += b
                if b != 255: break
        ip += lit; op += lit
        if ip >= n: break
        off = blk[ip] | blk[ip+1] << 8; ip += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = blk[ip]; ip += 1; ml += b
                if b != 255: break
        ml += 4
        D.append((op, off, ml)); op += ml
    mops = [d[0] for d in D]; ends = [d[0] + d[2] for d in D]
    def deps(idx, s_lo, s_hi):
        if s_hi <= s_lo: return (1, 0)
        i = bisect.bisect_right(ends, s_lo, 0, idx)      # first e with end > s_lo
        j = bisect.bisect_left(mops, s_hi, 0, idx) - 1    # last e with mop < s_hi
        return (i, j)
    def run(redirect_rounds):
        src = [d[0] - d[1] for d in D]
        rng = []
        for idx, (mop, off, ml) in enumerate(D):
            s = src[idx]; rng.append(deps(idx, max(s, 0), min(s + ml, mop)))
        red = 0
        for _ in range(redirect_rounds):
            nsrc = list(src); nrng = list(rng); ch = 0
            for idx, (mop, off, ml) in enumerate(D):
                i, j = rng[idx]
                if i == j and ml <= 16 and off >= ml:
                    e = i; emop, eoff, eml = D[e]
                    s = src[idx]
                    # source fully inside dest of e (as currently tracked: e copies from src[e], non-overlapping)
                    if s >= emop and s + ml <= emop + eml and (emop - src[e]) >= eml:
                        nsrc[idx] = src[e] + (s - emop); nrng[idx] = rng[e]; ch += 1
            src, rng = nsrc, nrng; red += ch
            if not ch: break
        depth = [0] * len(D)
        for idx in range(len(D)):
            i, j = rng[idx]
            depth[idx] = 1 + (max(depth[i:j+1]) if i <= j else 0)
        return max(depth), float(np.mean(depth)), red
    res.append((len(D), run(0), run(1), run(2), run(4), run(12)))
for r in res[:24]: print(r)
