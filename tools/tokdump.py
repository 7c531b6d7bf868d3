"""Debug aid: decode a raw deflate stream into its blocks and LZ77 tokens (position, length, distance)."""
LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
ORDER = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]


class Bits:
    def __init__(self, b): self.b = b; self.p = 0
    def get(self, n):
        v = 0
        for i in range(n):
            v |= ((self.b[self.p >> 3] >> (self.p & 7)) & 1) << i; self.p += 1
        return v


def table(lengths):
    codes = {}; code = 0
    for bits in range(1, 16):
        for s, l in enumerate(lengths):
            if l == bits:
                codes[(bits, code)] = s; code += 1
        code <<= 1
    return codes


def sym(br, tab):
    code = 0
    for bits in range(1, 16):
        code = (code << 1) | br.get(1)
        if (bits, code) in tab: return tab[(bits, code)]
    raise ValueError("bad code at bit %d" % br.p)


def tokens(raw):
    """-> list of blocks: dict(btype, bit_start, pos, toks=[(pos, len, dist) | (pos, 0, byte)])"""
    br = Bits(raw); out = []; pos = 0
    while True:
        start = br.p; final = br.get(1); bt = br.get(2); blk = dict(btype=bt, bit_start=start, pos=pos, toks=[], final=final)
        if bt == 0:
            br.p = (br.p + 7) & ~7; n = br.get(16); br.get(16)
            blk["stored"] = n; br.p += 8 * n; pos += n
        else:
            if bt == 1:
                ll = table([8]*144 + [9]*112 + [7]*24 + [8]*8); dd = table([5]*30)
            else:
                hl = br.get(5) + 257; hd = br.get(5) + 1; hc = br.get(4) + 4; cl = [0]*19
                for i in range(hc): cl[ORDER[i]] = br.get(3)
                ct = table(cl); ls = []
                while len(ls) < hl + hd:
                    s = sym(br, ct)
                    if s < 16: ls.append(s)
                    elif s == 16: ls += [ls[-1]] * (3 + br.get(2))
                    elif s == 17: ls += [0] * (3 + br.get(3))
                    else: ls += [0] * (11 + br.get(7))
                ll = table(ls[:hl]); dd = table(ls[hl:])
            while True:
                s = sym(br, ll)
                if s == 256: break
                if s < 256:
                    blk["toks"].append((pos, 0, s)); pos += 1
                else:
                    l = LBASE[s-257] + br.get(LEXT[s-257]); d = sym(br, dd); d = DBASE[d] + br.get(DEXT[d])
                    blk["toks"].append((pos, l, d)); pos += l
        out.append(blk)
        if final: break
    return out


def first_diff(a, b):
    ta = [t for blk in tokens(a) for t in blk["toks"]]; tb = [t for blk in tokens(b) for t in blk["toks"]]
    for i, (x, y) in enumerate(zip(ta, tb)):
        if x != y: return i, x, y
    return None
