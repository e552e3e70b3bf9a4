"""One-off source tool (round 5): resolves the compile-time A/B switches that lost their A/B and are never built --
keeps the branch of the shipped configuration, drops the other and the directive lines.  Conditionals on any other symbol are left alone.
    python scripts/resolve_knobs.py file ...
"""
import re
import sys

UNDEF = {'K3_NO_UNIFORM', 'K3_NO_COMPACT', 'K3_NO_INCR_GU', 'DN_STAMPS', 'DN_TRTRI_STAMPS'}
VALUES = {'DN_NEWTON': 2, 'K3_LOW_MODE': 1, 'DN_WAVES': 8, 'K3_GONDZIO': 3}


def decide(line):
    """True / False when the directive is about a resolved symbol, None otherwise"""
    m = re.match(r'\s*#\s*ifdef\s+(\w+)', line)
    if m:
        return False if m.group(1) in UNDEF else None
    m = re.match(r'\s*#\s*ifndef\s+(\w+)', line)
    if m:
        return True if m.group(1) in UNDEF else None
    m = re.match(r'\s*#\s*if\s+(\w+)\s*(>=|==|>)\s*(\d+)\s*$', line)
    if m and m.group(1) in VALUES:
        a, b = VALUES[m.group(1)], int(m.group(3))
        return {'>=': a >= b, '==': a == b, '>': a > b}[m.group(2)]
    return None


def resolve(text):
    out, stack = [], []          # stack entries: [resolved?, keep_now, parent_keep]
    for line in text.splitlines(keepends=True):
        keep = all(e[1] for e in stack if e[0])
        if re.match(r'\s*#\s*if', line):
            d = decide(line)
            if d is None:
                stack.append([False, True, keep])
                if keep:
                    out.append(line)
            else:
                stack.append([True, d, keep])
            continue
        if re.match(r'\s*#\s*else', line) and stack:
            if stack[-1][0]:
                stack[-1][1] = not stack[-1][1]
            elif all(e[1] for e in stack if e[0]):
                out.append(line)
            continue
        if re.match(r'\s*#\s*elif', line) and stack and stack[-1][0]:
            raise SystemExit('elif on a resolved symbol: not handled: ' + line)
        if re.match(r'\s*#\s*endif', line) and stack:
            e = stack.pop()
            if not e[0] and all(x[1] for x in stack if x[0]):
                out.append(line)
            continue
        if keep:
            out.append(line)
    assert not stack
    return ''.join(out)


for p in sys.argv[1:]:
    s = open(p).read()
    t = resolve(s)
    # '#ifndef X / #define X v / #endif' of the resolved parameters -> plain define
    t = re.sub(r'#ifndef (DN_NEWTON)\n(#define \1[^\n]*\n)#endif\n', r'\2', t)
    if t != s:
        open(p, 'w').write(t)
        print('resolved', p)
