#!/bin/bash
# GPU box: same-box A/B of the fp16-split 32-row slices encoder (enc_f16_slices.cuh) against the kernels it replaced
# (GNNCCA_DIAG=1 GNNCCA_GEMM_SLICES_MAX=0) and a few caps.  usage: bash tools/ab_slices.sh [quick]
run() {
  python bench.py --steps 20 --warmup 5 --no-train --no-config4 --no-scale-probe --no-cpu-baseline --no-terrace 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', 'headline %.2f' % (d['ms_per_step']*1e3), 'enc %.2f+%.2f' % (d['kernels_us']['enc_gemm'], d['kernels_us']['enc_tail']), ' | '.join('%s %.2f (enc %.1f+%.1f)' % (k[:14], v['ms_per_step']*1e3, v['kernels_us']['enc_gemm'], v['kernels_us']['enc_tail']) for k,v in d['configs'].items()))"
}
enc() {
  python tools/time_encoder.py 256 512 1024 1229 2048 3000 4000 2>/dev/null | python -c "
import json,sys
print('$1 ring ', ' '.join('%s: %.1f+%.1f' % (k, v['enc_gemm'][0], v['enc_tail'][0]) for k,v in json.loads(sys.stdin.read()).items()))"
  python tools/time_encoder.py 1024 2048 4000 --dense 64 2>/dev/null | python -c "
import json,sys
print('$1 dense', ' '.join('%s: %.1f+%.1f' % (k, v['enc_gemm'][0], v['enc_tail'][0]) for k,v in json.loads(sys.stdin.read()).items()))"
}
unset GNNCCA_DIAG; enc default; run default
export GNNCCA_DIAG=1
GNNCCA_GEMM_SLICES_MAX=0 enc replaced; GNNCCA_GEMM_SLICES_MAX=0 run replaced
if [ "$1" != quick ]; then
GNNCCA_GEMM_SLICES_NKS=16 run nks16
GNNCCA_GEMM_SLICES_WGS=512 enc wg512; GNNCCA_GEMM_SLICES_WGS=512 run wg512
fi
unset GNNCCA_DIAG; run default
