#!/usr/bin/env python3
"""Minimal reader for R `save()` files (.rda, XDR serialisation v2/v3).

Parses the byte stream only -- nothing from the file is executed (no pickle, no
R).  Used in the build container to lift *data columns* out of the reference's
bundled datasets (reference data/*.rda) into small text fixtures
under tests/golden/.  The reference does not travel to the GPU box; the
fixtures do.

    python tools/rda_reader.py /root/reference/data/evp_peparray.rda PROBE_SEQUENCE \
        > tests/golden/evp_peparray_probe_sequence.txt
"""
import bz2, gzip, lzma, struct, sys


class _Stream:
    def __init__(self, b):
        self.b, self.p = b, 0

    def take(self, n):
        v = self.b[self.p:self.p + n]
        if len(v) != n:
            raise EOFError
        self.p += n
        return v

    def i32(self):
        return struct.unpack(">i", self.take(4))[0]

    def f64(self):
        return struct.unpack(">d", self.take(8))[0]


NILVALUE, GLOBALENV, EMPTYENV, BASEENV, MISSINGARG, UNBOUND, BASENS = 254, 253, 242, 241, 251, 252, 247
REFSXP, NAMESPACESXP, PACKAGESXP, PERSISTSXP, ALTREP = 255, 249, 250, 248, 238
ATTRLISTSXP, ATTRLANGSXP = 239, 240


class Reader:
    def __init__(self, raw):
        if raw[:3] == b"BZh":
            raw = bz2.decompress(raw)
        elif raw[:2] == b"\x1f\x8b":
            raw = gzip.decompress(raw)
        elif raw[:6] == b"\xfd7zXZ\x00":
            raw = lzma.decompress(raw)
        if raw[:5] not in (b"RDX2\n", b"RDX3\n"):
            raise ValueError("not an RDX2/3 file")
        self.s = _Stream(raw)
        self.s.take(5)
        if self.s.take(2) != b"X\n":
            raise ValueError("only XDR format supported")
        ver = self.s.i32()
        self.s.i32(); self.s.i32()
        if ver == 3:
            n = self.s.i32()
            self.s.take(n)  # native encoding name
        self.refs = []

    def _len(self):
        n = self.s.i32()
        if n == -1:
            hi, lo = self.s.i32(), self.s.i32()
            n = (hi << 32) + lo
        return n

    def item(self):
        flags = self.s.i32()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)
        if t == NILVALUE:
            return None
        if t in (GLOBALENV, EMPTYENV, BASEENV, MISSINGARG, UNBOUND, BASENS):
            return ("env", t)
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.s.i32()
            return self.refs[idx - 1]
        if t == 1:  # SYMSXP
            name = self.item()
            sym = ("sym", name)
            self.refs.append(sym)
            return sym
        if t in (NAMESPACESXP, PACKAGESXP, PERSISTSXP):
            self.s.i32()
            n = self.s.i32()
            v = ("ns", [self.item() for _ in range(n)])
            self.refs.append(v)
            return v
        if t == 4:  # ENVSXP
            self.s.i32()
            env = {"_env": True}
            self.refs.append(env)
            env["enclos"], env["frame"], env["hashtab"], env["attr"] = (self.item() for _ in range(4))
            return env
        if t in (2, 6, 5, 17, ATTRLISTSXP, ATTRLANGSXP):  # pairlist-like
            out = []
            while True:
                attr = self.item() if (has_attr or t in (ATTRLISTSXP, ATTRLANGSXP)) else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag[1].decode() if tag else None, car))
                # cdr
                nf = self.s.i32()
                nt = nf & 0xFF
                if nt == NILVALUE:
                    break
                if nt not in (2, 6, ATTRLISTSXP, ATTRLANGSXP):
                    self.s.p -= 4
                    out.append(("_cdr", self.item()))
                    break
                t, has_attr, has_tag = nt, bool(nf & 0x200), bool(nf & 0x400)
            return ("pairlist", out)
        if t == 9:  # CHARSXP
            n = self.s.i32()
            return None if n == -1 else self.s.take(n)
        if t == 10 or t == 13:  # LGL / INT
            n = self._len()
            v = list(struct.unpack(">%di" % n, self.s.take(4 * n)))
        elif t == 14:
            n = self._len()
            v = list(struct.unpack(">%dd" % n, self.s.take(8 * n)))
        elif t == 16:  # STRSXP
            n = self._len()
            v = [self.item() for _ in range(n)]
        elif t in (19, 20):  # VECSXP / EXPRSXP
            n = self._len()
            v = [self.item() for _ in range(n)]
        elif t == 24:  # RAWSXP
            n = self._len()
            v = self.s.take(n)
        elif t == ALTREP:
            info, state, attr = self.item(), self.item(), self.item()
            return ("altrep", info, state, attr)
        else:
            raise NotImplementedError("SEXP type %d at %d" % (t, self.s.p))
        attrs = {}
        if has_attr:
            a = self.item()
            if a:
                attrs = {k: val for k, val in a[1]}
        return {"v": v, "attr": attrs, "type": t}

    def toplevel(self):
        top = self.item()  # a pairlist of (name, value)
        return {k: v for k, v in top[1]}


def _expand_altrep(x):
    # compact_intseq / deferred strings are not needed for the columns we lift
    return x


def column(path, colname, objname=None):
    objs = Reader(open(path, "rb").read()).toplevel()
    if objname is None:
        objname = next(iter(objs))
    df = objs[objname]
    names = [b.decode() for b in df["attr"]["names"]["v"]]
    col = df["v"][names.index(colname)]
    if isinstance(col, tuple):
        raise NotImplementedError("ALTREP column")
    if col["type"] == 16:
        return [None if b is None else b.decode("latin-1") for b in col["v"]]
    if "levels" in col["attr"]:  # factor
        lv = [b.decode("latin-1") for b in col["attr"]["levels"]["v"]]
        return [lv[i - 1] for i in col["v"]]
    return col["v"]


if __name__ == "__main__":
    vals = column(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    for v in vals:
        print("" if v is None else v)
