"""Minimal reader for R `.rda` / `.rds` files (XDR serialisation, format 2/3).

TEST INFRASTRUCTURE ONLY.  R is not installed in the build image, so the
reference's bundled fixtures (`/root/reference/data/*.rda`) are decoded with
this reader by `tests/golden/make_golden.py`, which commits the extracted
arrays under `tests/golden/`.  Nothing on the product path imports this.

Format notes (R Internals, "Serialization Formats"): gzip/xz/bzip2 container,
magic `RDX2\n`/`RDX3\n` for save() files, then `X\n` (XDR big-endian), three
int32 (version, writer, min-reader), v3 adds the native encoding string.  Every
item starts with an int32 flag word: type = flags & 0xff, 0x100 = object bit,
0x200 = has attributes, 0x400 = has tag.
"""
from __future__ import annotations

import bz2
import gzip
import lzma
import struct

import numpy as np

NILVALUE_SXP = 254
REFSXP = 255
GLOBALENV_SXP = 253
UNBOUNDVALUE_SXP = 252
MISSINGARG_SXP = 251
BASENAMESPACE_SXP = 250
NAMESPACESXP = 249
PACKAGESXP = 248
PERSISTSXP = 247
EMPTYENV_SXP = 242
BASEENV_SXP = 241
ATTRLANGSXP = 240
ATTRLISTSXP = 239
ALTREP_SXP = 238


class RObject:
    """A decoded R value that carries attributes (S4 objects: attrs == slots)."""

    def __init__(self, value, attrs=None, kind=""):
        self.value = value
        self.attrs = attrs or {}
        self.kind = kind

    def __repr__(self):
        return f"RObject(kind={self.kind!r}, attrs={list(self.attrs)})"

    def __getitem__(self, key):
        if isinstance(self.value, dict):
            return self.value[key]
        return self.attrs[key]


class _Reader:
    def __init__(self, buf: bytes):
        self.buf = buf
        self.pos = 0
        self.refs = []

    def _int(self) -> int:
        v = struct.unpack_from(">i", self.buf, self.pos)[0]
        self.pos += 4
        return v

    def _length(self) -> int:
        n = self._int()
        if n == -1:
            hi = self._int()
            lo = self._int()
            n = (hi << 32) + (lo & 0xFFFFFFFF)
        return n

    def _bytes(self, n: int) -> bytes:
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b

    def header(self):
        if self.buf[self.pos:self.pos + 5] in (b"RDX2\n", b"RDX3\n"):
            self.pos += 5
        fmt = self._bytes(2)
        if fmt != b"X\n":
            raise ValueError(f"only XDR serialisation supported, got {fmt!r}")
        version = self._int()
        self._int()
        self._int()
        if version == 3:
            n = self._int()
            self._bytes(n)
        elif version != 2:
            raise ValueError(f"unsupported serialisation version {version}")

    def _attrs(self):
        a = self.item()
        return a if isinstance(a, dict) else {}

    def item(self):
        flags = self._int()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)

        if t == NILVALUE_SXP:
            return None
        if t in (GLOBALENV_SXP, EMPTYENV_SXP, BASEENV_SXP, BASENAMESPACE_SXP,
                 UNBOUNDVALUE_SXP, MISSINGARG_SXP):
            return RObject(None, kind=f"special:{t}")
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self._int()
            return self.refs[idx - 1]
        if t == 1:  # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if t in (NAMESPACESXP, PACKAGESXP, PERSISTSXP):
            self._int()
            n = self._int()
            info = [self.item() for _ in range(n)]
            obj = RObject(info, kind=f"nsref:{t}")
            self.refs.append(obj)
            return obj
        if t == 4:  # ENVSXP
            self._int()  # locked
            obj = RObject({}, kind="env")
            self.refs.append(obj)
            self.item()  # enclos
            obj.value["frame"] = self.item()
            self.item()  # hashtab
            self.item()  # attrib
            return obj
        if t in (2, 6, 5, 17, ATTRLANGSXP, ATTRLISTSXP):  # pairlist-like
            # iterative over cdr to survive long lists
            out = {}
            order = 0
            first_attrs = None
            while True:
                attrs = self._attrs() if has_attr else None
                if first_attrs is None:
                    first_attrs = attrs
                tag = self.item() if has_tag else None
                car = self.item()
                key = tag if isinstance(tag, str) else f"__{order}"
                if key in out:
                    key = f"{key}__{order}"
                out[key] = car
                order += 1
                # peek next flags for cdr
                nflags = self._int()
                nt = nflags & 0xFF
                if nt in (2, 6, 5, 17, ATTRLANGSXP, ATTRLISTSXP):
                    has_attr = bool(nflags & 0x200)
                    has_tag = bool(nflags & 0x400)
                    continue
                self.pos -= 4
                tail = self.item()
                if tail is not None:
                    out[f"__tail{order}"] = tail
                break
            return out
        if t == 3:  # CLOSXP
            attrs = self._attrs() if has_attr else {}
            env = self.item() if has_tag else None
            formals = self.item()
            body = self.item()
            return RObject({"env": env, "formals": formals, "body": body}, attrs, "closure")
        if t in (7, 8):  # SPECIALSXP / BUILTINSXP
            n = self._int()
            return RObject(self._bytes(n).decode("latin-1"), kind="builtin")
        if t == 9:  # CHARSXP
            n = self._int()
            if n == -1:
                return None
            return self._bytes(n).decode("utf-8", errors="replace")
        if t in (10, 13):  # LGLSXP / INTSXP
            n = self._length()
            v = np.frombuffer(self.buf, dtype=">i4", count=n, offset=self.pos).astype(np.int32)
            self.pos += 4 * n
            return self._finish(v, has_attr, "lgl" if t == 10 else "int")
        if t == 14:  # REALSXP
            n = self._length()
            v = np.frombuffer(self.buf, dtype=">f8", count=n, offset=self.pos).astype(np.float64)
            self.pos += 8 * n
            return self._finish(v, has_attr, "real")
        if t == 15:  # CPLXSXP
            n = self._length()
            v = np.frombuffer(self.buf, dtype=">c16", count=n, offset=self.pos).astype(np.complex128)
            self.pos += 16 * n
            return self._finish(v, has_attr, "cplx")
        if t == 16:  # STRSXP
            n = self._length()
            v = [self.item() for _ in range(n)]
            return self._finish(v, has_attr, "str")
        if t in (19, 20):  # VECSXP / EXPRSXP
            n = self._length()
            v = [self.item() for _ in range(n)]
            return self._finish(v, has_attr, "list")
        if t == 24:  # RAWSXP
            n = self._length()
            v = np.frombuffer(self.buf, dtype=np.uint8, count=n, offset=self.pos).copy()
            self.pos += n
            return self._finish(v, has_attr, "raw")
        if t == 25:  # S4SXP: slots are the attributes
            attrs = self._attrs() if has_attr else {}
            return RObject(None, attrs, "S4")
        if t == 21:  # BCODESXP
            raise ValueError("bytecode objects are not supported")
        if t == 22:  # EXTPTRSXP
            obj = RObject(None, kind="extptr")
            self.refs.append(obj)
            self.item()
            self.item()
            if has_attr:
                self._attrs()
            return obj
        if t == ALTREP_SXP:
            info = self.item()
            state = self.item()
            self.item()  # attr
            cls = info.get("__0") if isinstance(info, dict) else None
            if cls == "compact_intseq":
                n, start, step = (int(x) for x in state)
                return (start + step * np.arange(n)).astype(np.int32)
            if cls == "compact_realseq":
                n, start, step = state
                return start + step * np.arange(int(n), dtype=np.float64)
            if cls in ("deferred_string",):
                return state
            if cls in ("wrap_real", "wrap_integer", "wrap_string", "wrap_logical"):
                return state[0] if isinstance(state, list) else state
            raise ValueError(f"unsupported ALTREP class {cls}")
        raise ValueError(f"unsupported SEXP type {t} at byte {self.pos}")

    def _finish(self, v, has_attr, kind):
        if not has_attr:
            return v
        attrs = self._attrs()
        if kind == "list" and "names" in attrs and len(attrs["names"]) == len(v):
            d = dict(zip(attrs["names"], v))
            return RObject(d, attrs, "namedlist")
        if isinstance(v, np.ndarray) and "dim" in attrs and len(attrs["dim"]) == 2 \
                and set(attrs) <= {"dim", "dimnames"}:
            r, c = (int(x) for x in attrs["dim"])
            m = v.reshape(c, r).T  # column-major -> (rows, cols) view
            return RObject(m, attrs, "matrix")
        return RObject(v, attrs, kind)


def _decompress(raw: bytes) -> bytes:
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    if raw[:6] == b"\xfd7zXZ\x00":
        return lzma.decompress(raw)
    if raw[:3] == b"BZh":
        return bz2.decompress(raw)
    return raw


def read_rda(path: str) -> dict:
    """Return {variable name: value} for a save()-style `.rda` file."""
    with open(path, "rb") as fh:
        buf = _decompress(fh.read())
    rd = _Reader(buf)
    rd.header()
    top = rd.item()
    if not isinstance(top, dict):
        raise ValueError("expected a pairlist at top level of .rda")
    return top


def as_matrix(obj) -> np.ndarray:
    """Matrix RObject -> C-contiguous (rows, cols) ndarray."""
    if isinstance(obj, RObject) and obj.kind == "matrix":
        return np.ascontiguousarray(obj.value)
    raise TypeError(f"not a matrix: {obj!r}")


def factor_codes(obj):
    """Factor RObject -> (0-based codes, levels)."""
    codes = np.asarray(obj.value, dtype=np.int64) - 1
    return codes, list(obj.attrs["levels"])
