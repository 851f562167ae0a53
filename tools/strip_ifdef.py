#!/usr/bin/env python3
"""strip_ifdef.py MACRO file... -- remove the `#ifdef MACRO` / `#if defined(MACRO) ...` blocks of a retired tuning macro from
sources (the `#else` branch, if any, stays); nested conditionals inside are handled.  Round 6: how the concluded
experiments (-DPIC1DP_TUNE_SUMS2) left the product sources."""
import re
import sys


def strip(text, macro):
    out, stack = [], []        # stack entries: [is_target, keeping]
    for line in text.split("\n"):
        st = line.strip()
        if re.match(r"#\s*if", st):
            target = bool(re.match(r"#\s*ifdef\s+%s\b" % macro, st)) or bool(re.match(r"#\s*if\s+defined\(%s\)" % macro, st))
            neg = bool(re.match(r"#\s*ifndef\s+%s\b" % macro, st))
            if target or neg:
                stack.append([True, neg])
                continue
            stack.append([False, True])
        elif re.match(r"#\s*else", st) and stack and stack[-1][0]:
            stack[-1][1] = not stack[-1][1]
            continue
        elif re.match(r"#\s*endif", st):
            top = stack.pop()
            if top[0]:
                continue
        if all(k for t, k in stack if t):
            out.append(line)
    assert not stack
    return "\n".join(out)


if __name__ == "__main__":
    macro = sys.argv[1]
    for path in sys.argv[2:]:
        s = open(path).read()
        t = strip(s, macro)
        if t != s:
            open(path, "w").write(t)
            print("stripped", macro, "from", path)
