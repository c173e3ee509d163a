"""`python -m mplan2vdl_amd.frontend DIR [flags] plan.mplan` -- the reference's `./tpchrun DIR plan`
(/root/reference/tpchrun:2-4): prints the VDL program for a MonetDB logical plan.  `-` reads the plan
from stdin (MainFuns.hs:142)."""
import argparse
import sys

from . import compile_plan, load_metadata


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m mplan2vdl_amd.frontend")
    ap.add_argument("metadata_dir")
    ap.add_argument("plan", nargs="?", default="-")
    ap.add_argument("--metadata", "--meta", action="store_true", dest="show_metadata", help="append ;; Metadata {...} to every line")
    ap.add_argument("-p", "--push-joins", action="store_true")
    ap.add_argument("-c", "--no-cleanup-passes", action="store_true")
    ap.add_argument("--use-cross-product", action="store_true")
    ap.add_argument("--goffset", type=int, default=0)
    ap.add_argument("--aggshuffle", action="store_true")
    ap.add_argument("--agghierarchical", action="store_true")
    ap.add_argument("-g", "--grainsize", type=int, default=8192)
    ap.add_argument("--vliteformat", action="store_true", help="lighter syntax (one value per vector), MainFuns.hs:70")
    ap.add_argument("--vdlformat", action="store_true", help="the default")
    a = ap.parse_intermixed_args(argv)
    strat = ("AggSerial",)
    if a.aggshuffle: strat = ("AggShuffle",)
    if a.agghierarchical: strat = ("AggHierarchical", a.grainsize.bit_length() - 1)
    cfg = load_metadata(a.metadata_dir, show_metadata=a.show_metadata, cross_product=a.use_cross_product,
                        gboffset=a.goffset, aggregation_strategy=strat, format="vlite" if a.vliteformat else "vdl")
    text = sys.stdin.read() if a.plan == "-" else open(a.plan).read()
    print(compile_plan(text, cfg, apply_passes=not a.no_cleanup_passes, push_joins=a.push_joins))


if __name__ == "__main__":
    main()
