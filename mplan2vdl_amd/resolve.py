"""Result decoding: the reference's resolve.py (/root/reference/resolve.py, Python 2) restated for Python 3.

Input: the executor's JSON reply {"results": {tmpN: {".<name>": [ints] | null}}, "timings": {...}} and the
catalog's dictionary.csv (table, column, "string", code).  Output names of the form out__table__col
(resolve.py:64-78) are dictionary-decoded with that column's codes; rows are padded with '-' to the longest
column (resolve.py:108-119) and written as CSV.

    tools/tpchrun DIR plan | vdlrun --rows N | python -m mplan2vdl_amd.resolve DIR/dictionary.csv
"""
import csv
import json
import sys


def load_dictionary(path):
    resolver = {}
    with open(path, newline="") as f:
        for tab, col, strng, code in csv.reader(f):
            resolver.setdefault("%s.%s" % (tab, col), {})[int(code)] = strng
    return resolver


def decode(reply, resolver, warn=None):
    """-> (names, rows).  Mirrors resolve.py:52-119."""
    warn = warn or (lambda msg: None)
    if "results" not in reply:
        raise ValueError("no results available")
    cols = []
    for res in reply["results"].values():
        if len(res) != 1:
            warn("unexpected: more than one path in output")
        k, vals = next(iter(res.items()))
        if vals is None:
            warn("WARNING: full column is null... continuing %s" % k)
            vals = []
        names = k.split("__")
        if len(names) != 3:
            warn(("origin not know for column %s" if len(names) < 3 else "name with more than 2 parts %s") % k)
            cols.append((k, list(vals)))
            continue
        dictname = ".".join(names[1:])
        if dictname not in resolver:
            warn("dictionary not found for %s" % dictname)
            cols.append((k, list(vals)))
            continue
        dec = resolver[dictname]
        out = []
        for v in vals:
            if v not in dec:
                warn("decoder has no mapping for: %s %s" % (dictname, v))
            out.append(dec.get(v, v))
        cols.append((names[0], out))
    width = max((len(v) for _, v in cols), default=0)
    padded = [v + ["-"] * (width - len(v)) for _, v in cols]
    return [n for n, _ in cols], [list(r) for r in zip(*padded)]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        sys.stderr.write("usage: python -m mplan2vdl_amd.resolve dictionary.csv < reply.json\n")
        return 2
    names, rows = decode(json.load(sys.stdin), load_dictionary(argv[0]), warn=lambda m: sys.stderr.write(m + "\n"))
    w = csv.writer(sys.stdout)
    w.writerow(names)
    w.writerows(rows)
    return 0


if __name__ == "__main__":
    sys.exit(main())
