#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_abl
mkdir -p $OUT
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --workload ${WL:-mg94_64x10k} --steps 100 --warmup 10 --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  python - $tag $OUT/$tag.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    print(f"{tag:34s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:8.1f} us  prune {r['kernel_ms']*1e3:8.1f} us  frac {r['frac']:.3f}")
except Exception as e:
    print(f"{tag:34s} FAILED ({e})")
PY
}
run m12 HYPHY_HIP_CHAIN_M=12
run m12_no_deposit_stores HYPHY_HIP_CHAIN_M=12 HYPHY_HIP_ABLATE=256
run m12_no_deposit_loads HYPHY_HIP_CHAIN_M=12 HYPHY_HIP_ABLATE=512
run m12_free_joins HYPHY_HIP_CHAIN_M=12 HYPHY_HIP_ABLATE=768
run m5_free_joins HYPHY_HIP_CHAIN_M=5 HYPHY_HIP_ABLATE=768
run m3_free_joins HYPHY_HIP_CHAIN_M=3 HYPHY_HIP_ABLATE=768
run m12_noparking HYPHY_HIP_CHAIN_M=12 HYPHY_HIP_SLOTS=2
run m12_occ3_noparking HYPHY_HIP_CHAIN_M=12 HYPHY_HIP_SLOTS=2 HYPHY_HIP_LIB=$GRAFT_REPO_ROOT/hyphy_amd/lib_occ3/libhyphy_hip.so
run m8_occ3_noparking HYPHY_HIP_CHAIN_M=8 HYPHY_HIP_SLOTS=2 HYPHY_HIP_LIB=$GRAFT_REPO_ROOT/hyphy_amd/lib_occ3/libhyphy_hip.so
run m5_occ3_noparking HYPHY_HIP_CHAIN_M=5 HYPHY_HIP_SLOTS=2 HYPHY_HIP_LIB=$GRAFT_REPO_ROOT/hyphy_amd/lib_occ3/libhyphy_hip.so
