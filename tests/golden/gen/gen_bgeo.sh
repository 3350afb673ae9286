#!/usr/bin/env bash
# Regenerates tests/golden/g10_points.f32 and g10_partio.bgeo: a fixed point set written by the reference's own partio
# library (Externals/partio/core/*.cpp + io/*.cpp compiled where they lie, no zlib: uncompressed .bgeo needs none).
# Runs only where /root/reference is mounted; tests read the committed files.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$(cd "$HERE/.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
P="$REF/Externals/partio"
g++ -std=c++11 -O1 -w -I"$P" "$HERE/gen_bgeo.cpp" "$P"/core/*.cpp "$P"/io/*.cpp -o "$TMP/gen_bgeo" -lpthread
python3 - "$OUT/g10_points.f32" <<'PY'
import sys, numpy as np
# 1000 deterministic points: lattice samples, denormal-free extremes, negatives, values that need all 4 bytes
i = np.arange(1000, dtype=np.float64)
pts = np.stack([np.sin(0.37 * i) * 0.5 + 0.5, (i * 0.6180339887) % 1.0, np.cos(1.1 * i) * 3.0], axis=1).astype(np.float32)
pts[0] = (0.0, 1.0, -1.0)
pts[1] = (np.float32(1e-30), np.float32(3.4e38), np.float32(-2.5e-7))
pts[2] = (0.25, 0.5, 0.75)
pts.tofile(sys.argv[1])
PY
"$TMP/gen_bgeo" "$OUT/g10_points.f32" "$TMP/g10_partio.bgeo"
cp "$TMP/g10_partio.bgeo" "$OUT/g10_partio.bgeo"
ls -l "$OUT/g10_points.f32" "$OUT/g10_partio.bgeo"
