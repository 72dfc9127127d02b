# GPU box: ablation matrix of the fp16-split encoder GEMM (enc_f16.cuh) at N = 65 536, same box, one process per arm.
# Needs the twin build: python gnn-cca_amd/build.py --f16-ablations   (GNNCCA_GEMM_F16_DIAG bits: 1 no MFMA, 2 x from L2, 4 no fp16 split)
L=$PWD/gnn-cca_amd/lib/libgnncca_mpn_f16abl.so
for rep in 1 2; do
for d in 0 1 2 3 4 5 6 7; do
  echo "--- diag $d"
  GNNCCA_DIAG=1 GNNCCA_LIB=$L GNNCCA_GEMM_F16_DIAG=$d python3 tools/time_encoder.py 65536 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:v['enc_gemm'] for k,v in d.items()})"
done; done
