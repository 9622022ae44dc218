"""Bit-level model of how k_spiht_decode (ebcc_amd/csrc/residual_spiht.hip) cuts the SPIHT stream into list entries,
checked against the sequential parse of the reference decoder (/root/reference/src/spiht/spiht_re.c:331-343 LIP pass,
:346-410 LIS pass).  The kernel's arithmetic is restated here in Python integers so that the two tricks it rests on are
pinned on the CPU as well as by the GPU parity tests:

* LIP pass: an entry is one bit, or two when the first is set.  The bits that are second bits ("signs") of the next 64
  stream bits have a closed form (the odd / even run-of-ones carry argument of simdjson's backslash scanner);
* LIS pass: a chunk is the entries that start in the first 55 of the next 64 stream bits; what a type-A entry would take
  is worked out per bit position (together with its children's significance / sign flags), a walk from set bit to set
  bit adds the lengths up (a run of zero bits is that many one-bit entries whatever their types; a guard bit at position
  55 ends the walk), and the list's end is cut afterwards.
"""
import random

M = (1 << 64) - 1
EVEN = 0x5555555555555555


def brev64(x):
    return int(format(x, "064b")[::-1], 2)


def stream64(bits, pos):
    """the kernel's stream64(): 64 bits from `pos`, first bit in bit 63; past the end reads as 0 (bitio.h:60-63)"""
    v = 0
    for i in range(64):
        v = (v << 1) | (bits[pos + i] if pos + i < len(bits) else 0)
    return v


def sign_bits(x):
    """bits of x (bit i = stream bit i, bit 0 an entry start) that are the second bit of a two-bit entry"""
    follows = (x << 1) & M
    odd_starts = x & ~EVEN & ~follows & M
    return (EVEN ^ (((odd_starts + x) & M) << 1)) & follows & M


def sign_bits_sequential(x):
    e, i = 0, 0
    while i < 64:
        if (x >> i) & 1:
            if i + 1 < 64:
                e |= 1 << (i + 1)
            i += 2
        else:
            i += 1
    return e


def test_sign_bits_closed_form():
    rng = random.Random(1)
    cases = [0, M, EVEN, (EVEN << 1) & M, 1 << 63, 3 << 62, 1, 3, 7]
    for t in range(100000):
        x = rng.getrandbits(64)
        if t % 3 == 0:
            x |= rng.getrandbits(64)
        if t % 5 == 0:
            x &= rng.getrandbits(64)
        cases.append(x)
    for x in cases:
        assert sign_bits(x) == sign_bits_sequential(x), hex(x)


def lip_sequential(bits, n):
    out, pos = [], 0
    for _ in range(n):
        out.append(pos)
        pos += 1 + (bits[pos] if pos < len(bits) else 0)
    return out, pos


def lip_chunked(bits, n):
    """the kernel's LIP loop: returns the entry start offsets and the bits consumed"""
    out, base, cnt = [], 0, 0
    while base < n:
        x = brev64(stream64(bits, cnt))
        starts = ~sign_bits(x) & M
        limit = 63 if (starts & x) >> 63 else 64          # a significant entry at bit 63 waits for the next chunk
        n_rem = n - base
        mover = 0
        for lane in range(64):
            idx = bin(starts & ((1 << lane) - 1)).count("1")
            if (starts >> lane) & 1 and lane < limit and idx >= n_rem:
                mover |= 1 << lane
        if mover:
            limit = (mover & -mover).bit_length() - 1
        valid = [lane for lane in range(64) if (starts >> lane) & 1 and lane < limit]
        assert valid, "a chunk always makes progress"
        out += [cnt + lane for lane in valid]
        cnt += limit
        base += len(valid)
    return out, cnt


def test_lip_chunks_match_sequential_parse():
    rng = random.Random(3)
    for _ in range(1500):
        density = rng.random()
        bits = [1 if rng.random() < density else 0 for _ in range(rng.randint(1, 400))]
        n = rng.randint(1, 150)
        assert lip_sequential(bits, n) == lip_chunked(bits, n)


def lis_sequential(bits, types):
    out, pos = [], 0
    for is_b in types:
        out.append(pos)
        sb = bits[pos]
        pos += 1
        if not is_b and sb:                                # :360-372: four children, each one bit or bit + sign
            for _ in range(4):
                pos += 1 + bits[pos]
    return out, pos


def ctz(x):
    return (x & -x).bit_length() - 1


def lis_chunked(bits, types):
    """the kernel's LIS loop: entry starts, bits consumed, and for every type-A entry on a set bit the children's
    (significant, negative) flags its lane read ahead (wsig / wneg)"""
    out, flags, base, cnt, n = [], [], 0, 0, len(types)
    low55 = (1 << 55) - 1
    while base < n:
        m = min(64, n - base)
        x = stream64(bits, cnt)
        typemask = sum(1 << i for i in range(m) if types[base + i])
        len_a, wsig, wneg = [], [], []
        for lane in range(64):                             # every lane: what a type-A entry at its bit would take
            wl = ((x << lane) & M) >> 32                   # the 32 stream bits from this lane's position on, first in bit 31
            la, sg, ng = 1, 0, 0
            if lane < 55 and wl >> 31:
                q = 1
                for k in range(4):
                    sb = (wl >> (31 - q)) & 1
                    sg |= sb << k
                    ng |= (sb & (wl >> (30 - q))) << k
                    q += 1 + sb
                la = q
            len_a.append(la); wsig.append(sg); wneg.append(ng)
        # the walk from set bit to set bit, a guard bit at position 55
        xr = (brev64(x) & low55) | (1 << 55)
        starts, rel = 0, 0
        while True:
            z = ctz(xr >> rel)
            starts |= ((2 << z) - 1) << rel
            pos = rel + z
            rel = pos + (1 if (typemask >> (bin(starts).count("1") - 1)) & 1 else len_a[pos])
            if rel >= 55:
                break
        rel -= (starts >> 55) & 1
        starts &= low55
        n_ent = bin(starts).count("1")
        if n_ent > m:                                      # the list ends inside the chunk
            lanes = [lane for lane in range(64) if (starts >> lane) & 1]
            rel = lanes[m]
            starts &= (1 << rel) - 1
            n_ent = m
        assert 0 < rel <= 64
        i = 0
        for lane in range(64):
            if (starts >> lane) & 1:
                out.append(cnt + lane)
                if not types[base + i] and (x >> (63 - lane)) & 1:
                    flags.append((cnt + lane, wsig[lane], wneg[lane]))
                i += 1
        cnt += rel
        base += n_ent
    return out, cnt, flags


def lis_sequential_flags(bits, types):
    out, pos = [], 0
    for is_b in types:
        start = pos
        sb = bits[pos]
        pos += 1
        if not is_b and sb:
            sg = ng = 0
            for k in range(4):
                s_ = bits[pos]
                pos += 1
                if s_:
                    sg |= 1 << k
                    ng |= bits[pos] << k
                    pos += 1
            out.append((start, sg, ng))
    return out


def test_lis_chunks_match_sequential_parse():
    rng = random.Random(7)
    for t in range(1500):
        density = rng.random() if t % 4 else rng.choice([0.0, 0.02, 0.98, 1.0])
        bits = [1 if rng.random() < density else 0 for _ in range(3000)]
        types = [rng.random() < (0.5 if t % 3 else rng.choice([0.0, 1.0])) for _ in range(rng.randint(1, 200))]
        starts, used, flags = lis_chunked(bits, types)
        assert (starts, used) == lis_sequential(bits, types)
        assert flags == lis_sequential_flags(bits, types)
