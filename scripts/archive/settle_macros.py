#!/usr/bin/env python3
"""Settle A/B macros of a source file at their shipped values: `#if M` / `#if M == n` / `#else` blocks are resolved to the live arm, the
`#ifndef M / #define M v / #endif` default guards are dropped (the #define stays while the name is still used in code, unguarded).
    python scripts/settle_macros.py file M=v [M=v ...]        (function-like macros: give the name only, e.g. DIL_VW_WAVES)"""
import re
import sys


def main():
    path, vals = sys.argv[1], {}
    for a in sys.argv[2:]:
        k, _, v = a.partition("=")
        vals[k] = v if v != "" else None
    src = open(path).read().split("\n")
    out, stack = [], []          # stack entries: (kind, live, taken) kind 'settled' | 'other'
    i = 0

    def emitting():
        return all(live for kind, live, _ in stack if kind == "settled")

    def evaluate(expr):
        names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
        if not names or not names <= set(k for k, v in vals.items() if v is not None):
            return None
        e = expr
        for k in sorted(names, key=len, reverse=True):
            e = re.sub(r"\b%s\b" % k, vals[k], e)
        e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", " !=")
        return bool(eval(e))
    while i < len(src):
        line = src[i]
        m = re.match(r"\s*#\s*(ifndef|ifdef|if|elif|else|endif)\b(.*)", line)
        if not m:
            if emitting():
                out.append(line)
            i += 1
            continue
        d, rest = m.group(1), m.group(2).split("//")[0].strip()
        if d == "ifndef" and rest in vals:
            # default guard: drop the guard lines, keep what is inside
            stack.append(("guard", True, True))
            i += 1
            continue
        if d in ("if", "ifdef", "ifndef"):
            v = evaluate(rest) if d == "if" else None
            if v is None:
                stack.append(("other", True, True))
                if emitting():
                    out.append(line)
            else:
                stack.append(("settled", v, v))
        elif d in ("elif", "else"):
            kind, live, taken = stack[-1]
            if kind == "settled":
                if d == "else":
                    stack[-1] = (kind, not taken, True)
                else:
                    v = evaluate(rest)
                    assert v is not None, line
                    stack[-1] = (kind, (not taken) and v, taken or v)
            elif emitting():
                out.append(line)
        else:
            kind, _, _ = stack.pop()
            if kind == "other" and emitting():
                out.append(line)
        i += 1
    assert not stack
    text = "\n".join(out)
    # a settled macro's #define stays only while code still uses the name
    for k, v in vals.items():
        uses = len(re.findall(r"\b%s\b" % k, text))
        defs = len(re.findall(r"^#define %s\b" % k, text, flags=re.M))
        if uses == defs:
            text = re.sub(r"^#define %s\b.*\n" % k, "", text, flags=re.M)
    open(path, "w").write(text)


if __name__ == "__main__":
    main()
