run() { python tools/time_encoder.py 8192 --dense 128 --reps 20 2>/dev/null | python -c "
import json,sys
v=json.loads(sys.stdin.read())['8192']; print('$1', ' '.join('%s %.2f' % (k, x[0]) for k,x in v.items()), 'sum %.1f' % (v['plan'][0]+v['enc_gemm'][0]+3*v['step'][0]+v['step_last'][0]))"; }
unset GNNCCA_DIAG; run base
export GNNCCA_DIAG=1
GNNCCA_WPS=1 run wps1
GNNCCA_WPS=2 run wps2
GNNCCA_WPS=4 run wps4
GNNCCA_NPW=2 GNNCCA_NPW_MIN_N=8192 GNNCCA_NPW_MIN_N_FIRST=8192 run npw2
GNNCCA_NPW=2 GNNCCA_NPW_MIN_N=8192 GNNCCA_NPW_MIN_N_FIRST=100000 run npw2_later
GNNCCA_STEP_NT=1 run nt1
GNNCCA_NO_PAD=1 run nopad
unset GNNCCA_DIAG; run base
