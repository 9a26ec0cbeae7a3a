#!/bin/bash
# One command for whoever has the .NET 8 SDK (and python3 + numpy + g++ for the comparison):
#
#     tools/ReferenceDump/run_all.sh /path/to/YetAnotherConsoleGameEngine [frames = 3] [--large]
#
# 1. the scene files committed under tests/golden/reference/<name>/scene.ysc (configs 1, 2; config 3 at 320x90; the reduced config 4 and
#    config 5 scenes of the test suite) are rendered by the REFERENCE's own RaytraceRenderer.TryFlipAndBlit - the dump lands next to each
#    scene.ysc, where tests/test_reference_goldens.py finds it;
# 2. with --large also the five BASELINE configurations at full size (scene files generated here by tools/scene_file.py, 0.3 GB in all;
#    dumps under $YCGE_REFERENCE_GOLDENS or ./reference_dumps - too large to commit);
# 3. tools/compare_dump.py holds every dump to the oracle and prints the table of the README: exit status 0 = parity pinned.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"; REPO="$(cd "$HERE/../.." && pwd)"
REF="${1:?usage: run_all.sh <checkout of NullandKale/YetAnotherConsoleGameEngine> [frames] [--large]}"; FRAMES="${2:-3}"; LARGE="${3:-}"
command -v dotnet >/dev/null || { echo "dotnet not found: this step needs the .NET 8 SDK"; exit 2; }
cd "$HERE"
dotnet build -c Release -p:ReferenceRoot="$REF" >/dev/null
run() { echo "-- $1"; dotnet run --no-build -c Release -p:ReferenceRoot="$REF" -- "$1" "$2" "$FRAMES"; }
for d in "$REPO"/tests/golden/reference/*/; do
  [ -f "$d/scene.ysc" ] && run "$d/scene.ysc" "$d"
done
DUMPS=()
if [ "$LARGE" = "--large" ]; then
  OUT="${YCGE_REFERENCE_GOLDENS:-$REPO/reference_dumps}"; mkdir -p "$OUT"
  for spec in "1 config1" "2 config2" "3 config3" "4 config4" "5 config5" "5 config5_noon --t01 0.5"; do set -- $spec
    mkdir -p "$OUT/$2"; [ -f "$OUT/$2/scene.ysc" ] || python3 "$REPO/tools/scene_file.py" "$1" "$OUT/$2/scene.ysc" --gzip "${@:3}"
    run "$OUT/$2/scene.ysc" "$OUT/$2"; DUMPS+=("$OUT/$2")
  done
fi
python3 "$REPO/tools/compare_dump.py" "$REPO"/tests/golden/reference/*/ "${DUMPS[@]}"
