"""Line overlap of a product file with reference files (test tooling; reads /root/reference, so it runs in the build
container only): lines stripped of all whitespace, comments and docstring text dropped, >= 12 characters."""
import io
import sys
import tokenize


def lines(path):
    src = open(path, encoding="utf-8", errors="replace").read()
    drop = set()
    try:
        for tok in tokenize.generate_tokens(io.StringIO(src).readline):
            if tok.type == tokenize.COMMENT or (tok.type == tokenize.STRING and tok.line.strip().startswith(('"""', "'''", 'r"""'))):
                for ln in range(tok.start[0], tok.end[0] + 1):
                    if tok.type == tokenize.STRING:
                        drop.add(ln)
    except tokenize.TokenError:
        pass
    out = []
    for i, l in enumerate(src.splitlines(), 1):
        if i in drop:
            continue
        l = l.split("#")[0]
        l = "".join(l.split())
        if len(l) >= 12:
            out.append(l)
    return out


if __name__ == "__main__":
    mine = lines(sys.argv[1])
    tot = set()
    for ref in [a for a in sys.argv[2:] if a != "-v"]:
        r = lines(ref)
        rs = set(r)
        hit = [l for l in mine if l in rs]
        back = [l for l in r if l in set(mine)]
        tot |= set(hit)
        print("%-70s %3d of my %3d lines; %3d of its %3d lines" % (ref[-70:], len(hit), len(mine), len(back), len(r)))
    allhit = [l for l in mine if l in tot]
    print("total: %d of %d substantive lines (%.0f %%)" % (len(allhit), len(mine), 100.0 * len(allhit) / max(len(mine), 1)))
    if "-v" in sys.argv:
        for l in sorted(set(allhit)):
            print("   ", l)
