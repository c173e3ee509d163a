#!/bin/bash
# The reference's pipe (eval_query.sh:18-26) end to end without GHC, curl or a Voodoo server:
#   tools/tpchrun META plan | sed 's/;;.*//' | vdlrun --data COLDIR | python -m mplan2vdl_amd.resolve META/dictionary.csv
# over a synthetic catalog exported to COLDIR, for every plan given (default: all 15 that compile), and
# checked against the oracle's answer for the same program and columns.  Run from the repo root on a GPU box.
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
META=$ROOT/tests/golden/tpch10noorder
OUT=${E2E_OUT:-$ROOT/gpurun_out/e2e}
SCALE=${E2E_SCALE:-0.001}
mkdir -p "$OUT"
PLANS=${@:-01 03 04 05 06 09 10 11 12 14 15 16 18 19 20}
export PYTHONPATH=$ROOT:$ROOT/tests
fail=0
for q in $PLANS; do
  plan=$META/$q.sql.mplan
  "$ROOT/tools/tpchrun" "$META" "$plan" | sed 's/;;.*//' > "$OUT/$q.vdl"
  python3 - "$META" "$OUT/$q.vdl" "$OUT/cols_$q" "$SCALE" "$OUT/$q.want.json" <<'PY'
import json, sys
from mplan2vdl_amd import catalog, frontend
from helpers import oracle_run
meta, vdl, coldir, scale, want = sys.argv[1:6]
text = open(vdl).read()
cols = catalog.synth_columns(meta, frontend.load_metadata(meta), text, scale=float(scale))
catalog.export_columns(cols, coldir)
json.dump(oracle_run(text, cols), open(want, "w"))
PY
  "$ROOT/mplan2vdl_amd/bin/vdlrun" --data "$OUT/cols_$q" < "$OUT/$q.vdl" > "$OUT/$q.json"
  python3 -m mplan2vdl_amd.resolve "$META/dictionary.csv" < "$OUT/$q.json" > "$OUT/$q.csv"
  if python3 -c "import json,sys; a=json.load(open('$OUT/$q.json'))['results']; b=json.load(open('$OUT/$q.want.json')); sys.exit(0 if a==b else 1)"; then
    echo "Q$q ok: $(head -n 1 "$OUT/$q.csv" | cut -c1-100) ... $(($(wc -l < "$OUT/$q.csv") - 1)) row(s)"
  else
    echo "Q$q MISMATCH"; fail=1
  fi
  rm -rf "$OUT/cols_$q"
done
exit $fail
