#!/bin/bash
# two-stream schedule of the whole aggregation: which residency caps let the VALU-bound sealed-box kernels and the HBM-bound
# codec / share-gen kernels actually run side by side
cd "$(dirname "$0")/.."
run() { python tools/bench_full_aggregation.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.1f ms/tile %.2f Gelem/s ok=%s' % (d['ms_per_tile'], d['elements_per_s']/1e9, d['verified_reveal_equals_sum_of_secrets']))"; }
echo "serial:                 $(TILES=4 run)"
echo "two streams, no caps:   $(PIPELINE=1 TILES=6 run)"
for w in 1 2 4; do for sb in 0 2 4 6; do
  echo "two streams, wire cap $w, sbox cap $sb: $(PIPELINE=1 TILES=6 SDA_WIRE_WG_PER_CU=$w SDA_SBOX_WG_PER_CU=$sb run)"
done; done
