#!/usr/bin/env python3
"""Partial evaluation of preprocessor conditionals: a small `unifdef`.

    python tools/unifdef_lite.py FILE -DNAME=VALUE ... -UNAME ... -KNAME ... [-o OUT]

-DNAME=VALUE   NAME is defined with that value: `#if NAME`, `#if NAME == n`, `#ifdef NAME`, `#ifndef NAME` are decided and removed
-UNAME         NAME is not defined: `#ifdef NAME` blocks go, `#ifndef NAME` blocks stay (without the directive lines)
-KNAME         a knob with an in-file default, `#ifndef NAME / #define NAME v / #endif`: the guard lines go, the #define stays, and
               `#if NAME ...` is decided with v (the value of the #define found in the file)
Conditionals on other names are kept as they are.  `#elif` is only understood in chains whose every condition is decidable.
Used once in round 6 to take the experiment knobs of rounds 3-5 out of vf_kernels.h; the reverse of that edit is
tools/experiments/r06_kernel_laboratory.patch.
"""
import re
import sys


def main(argv):
    defs, undefs, knobs, out, path = {}, set(), set(), None, None
    it = iter(argv)
    for a in it:
        if a.startswith("-D"):
            k, _, v = a[2:].partition("=")
            defs[k] = v or "1"
        elif a.startswith("-U"):
            undefs.add(a[2:])
        elif a.startswith("-K"):
            knobs.add(a[2:])
        elif a == "-o":
            out = next(it)
        else:
            path = a
    lines = open(path).read().split("\n")
    # knob defaults
    for i, ln in enumerate(lines):
        m = re.match(r"\s*#\s*define\s+(\w+)\s+(\S+)", ln)
        if m and m.group(1) in knobs and m.group(1) not in defs:
            defs[m.group(1)] = m.group(2)

    def decide(directive, expr):
        """True / False, or None when the condition is not ours to decide"""
        expr = expr.split("//")[0].strip()
        if directive in ("ifdef", "ifndef"):
            name = expr.split()[0]
            if name in knobs and directive == "ifndef":
                return "knob"
            if name in defs and name not in knobs:
                return directive == "ifdef"
            if name in undefs:
                return directive == "ifndef"
            return None
        names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
        if not names or not names <= (set(defs) | undefs):
            return None
        e = re.sub(r"defined\s*\(?\s*(\w+)\s*\)?", lambda m: "1" if m.group(1) in defs else "0", expr)
        e = re.sub(r"[A-Za-z_]\w*", lambda m: defs.get(m.group(0), "0"), e)
        e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
        e = re.sub(r"(\d+)[uU]\b", r"\1", e)
        return bool(eval(e))  # noqa: S307 (own source file)

    res = []
    stack = []   # per open conditional: dict(kind: 'keep' | 'ours' | 'knob', taken: bool, active: bool, emit_parent: bool)
    emitting = True
    for ln in lines:
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", ln)
        if not m:
            if emitting:
                res.append(ln)
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("if", "ifdef", "ifndef"):
            v = decide(d, rest) if emitting else None
            if not emitting:
                stack.append({"kind": "dead", "parent": emitting})
                continue
            if v is None:
                stack.append({"kind": "keep", "parent": emitting})
                res.append(ln)
            elif v == "knob":
                stack.append({"kind": "knob", "parent": emitting})
            else:
                stack.append({"kind": "ours", "parent": emitting, "taken": v})
                emitting = v
        elif d in ("elif", "else"):
            top = stack[-1]
            if top["kind"] == "keep":
                res.append(ln)
            elif top["kind"] == "ours":
                if top["taken"]:
                    emitting = False
                elif d == "else":
                    emitting = top["parent"]; top["taken"] = True
                else:
                    v = decide("if", rest)
                    assert v is not None, f"undecidable #elif in a decided chain: {ln}"
                    emitting = top["parent"] and v; top["taken"] = v
            elif top["kind"] == "knob":
                raise SystemExit(f"#else in a knob guard: {ln}")
        else:
            top = stack.pop()
            if top["kind"] == "keep":
                res.append(ln)
            emitting = top["parent"]
    assert not stack
    text = "\n".join(res)
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1:])
