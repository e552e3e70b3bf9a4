"""Line-overlap of files in this repository with the reference's sources (the measure VERDICT r4 used: whitespace-normalised lines of at least 25
characters, comments and includes left out, against every source file of the reference).  Build container only: needs /root/reference.

    python tests/tools/overlap_check.py tests/cpp/*.cpp            # prints matched / counted lines per file
"""
import os
import re
import sys

REF = '/root/reference'
EXT = ('.cpp', '.h', '.hpp', '.cc', '.py', '.c', '.cu', '.hip')


def norm_lines(text):
    out = []
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    for ln in text.splitlines():
        ln = re.sub(r'//.*$', '', ln)
        ln = re.sub(r'\s+', '', ln)
        if len(ln) < 25 or ln.startswith('#include'):
            continue
        out.append(ln)
    return out


def reference_lines():
    ref = set()
    for d, _, files in os.walk(REF):
        if '/.git' in d:
            continue
        for f in files:
            if f.endswith(EXT):
                try:
                    ref.update(norm_lines(open(os.path.join(d, f), errors='ignore').read()))
                except OSError:
                    pass
    return ref


def overlap(path, ref):
    mine = norm_lines(open(path, errors='ignore').read())
    hit = sum(1 for ln in mine if ln in ref)
    return hit, len(mine)


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('reference not present')
    ref = reference_lines()
    for p in sys.argv[1:]:
        h, n = overlap(p, ref)
        print('%-60s %4d / %4d  %5.1f %%' % (p, h, n, 100.0 * h / max(n, 1)))
