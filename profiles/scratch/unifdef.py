#!/usr/bin/env python3
"""A small unifdef: resolve the preprocessor conditionals of a source file whose macros are decided (round 6's prune of the
measured-and-rejected variants), keep every other conditional as it is.
    unifdef.py file NAME=VALUE ... -UNAME ...   (in place)"""
import re
import sys

path = sys.argv[1]
known = {}
for a in sys.argv[2:]:
    if a.startswith("-U"):
        known[a[2:]] = None
    else:
        k, v = a.split("=")
        known[k] = int(v)


def evaluate(kind, expr):
    """True / False when decided, None when the line stays"""
    expr = expr.split("//")[0].strip()
    if kind == "ifdef":
        return (known[expr] is not None) if expr in known else None
    if kind == "ifndef":
        return (known[expr] is None) if expr in known else None
    names = set(re.findall(r"[A-Za-z_][A-Za-z_0-9]*", expr)) - {"defined"}
    if not names or not names <= set(known):
        return None
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if known[m.group(1)] is not None else "0", expr)
    e = re.sub(r"[A-Za-z_]\w*", lambda m: str(known[m.group(0)] or 0), e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
    return bool(eval(e))


out = []
stack = []   # entries: dict(decided=bool|None, taken=bool, emit=bool)
emit = True
lines = open(path).read().split("\n")
for ln in lines:
    m = re.match(r"^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$", ln)
    if not m:
        if emit:
            out.append(ln)
        continue
    kind, rest = m.group(1), m.group(2)
    if kind in ("if", "ifdef", "ifndef"):
        v = evaluate(kind, rest) if emit else None
        stack.append(dict(decided=v is not None, taken=bool(v), outer=emit, value=v))
        if v is None:
            if emit:
                out.append(ln)
        else:
            emit = emit and v
    elif kind == "elif":
        top = stack[-1]
        if not top["decided"]:
            if top["outer"]:
                out.append(ln)
        else:
            if top["taken"]:
                emit = False
            else:
                v = evaluate("if", rest)
                assert v is not None, ln
                top["taken"] = v
                emit = top["outer"] and v
    elif kind == "else":
        top = stack[-1]
        if not top["decided"]:
            if top["outer"]:
                out.append(ln)
        else:
            emit = top["outer"] and not top["taken"]
            top["taken"] = True
    else:
        top = stack.pop()
        if not top["decided"]:
            if top["outer"]:
                out.append(ln)
        emit = top["outer"]
assert not stack
open(path, "w").write("\n".join(out))
