#!/usr/bin/env python3
"""Formal-argument lists of R functions, read from R source text without R: `name <- function(a, b = default, ...)`.
Used twice: tests/golden/make_golden.py extracts the signatures of the reference's step functions into
tests/golden/r_step_function_signatures.json (names and order of the formals, and which carry a default -- API surface,
not source), and tests/test_host.py reads the R glue's `hip_*` functions the same way to compare them."""
import re


def _strip(src):
    """R source with comments removed and string literals blanked (lengths kept)."""
    out, q, i = [], None, 0
    while i < len(src):
        ch = src[i]
        if q:
            if ch == "\\" and i + 1 < len(src):
                out.append("  "); i += 2; continue
            if ch == q:
                q = None; out.append(ch)
            else:
                out.append(" " if ch != "\n" else "\n")
        elif ch in "\"'":
            q = ch; out.append(ch)
        elif ch == "#":
            while i < len(src) and src[i] != "\n":
                i += 1
            continue
        else:
            out.append(ch)
        i += 1
    return "".join(out)


def formals(src, name):
    """[(formal, has_default)] of `name <- function(...)` in R source `src`, or None."""
    s = _strip(src)
    m = re.search(r"(?m)^\s*" + re.escape(name) + r"\s*(?:<-|=)\s*function\s*\(", s)
    if not m:
        return None
    i, depth, start = m.end(), 1, m.end()
    parts = []
    while i < len(s) and depth:
        ch = s[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                parts.append(s[start:i])
        elif ch == "," and depth == 1:
            parts.append(s[start:i]); start = i + 1
        i += 1
    res = []
    for p in parts:
        p = p.strip()
        if not p:
            continue
        nm = p.split("=", 1)[0].strip()
        res.append((nm, "=" in p))
    return res
