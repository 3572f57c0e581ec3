for cfg in "GB_RING=1 GB_HEADS_FUSED=0" "GB_RING=0 GB_HEADS_FUSED=0" "GB_RING=1 GB_HEADS_FUSED=1" "GB_RING=1 GB_HEADS_FUSED=gd"; do
  env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || tail -3 gpurun_out/ab_tmp.err
  python - "$cfg" <<PY
import json, sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", d["ms_per_step"], "eager", d["ms_per_step_eager"], "gemm_cl", d["roofline"]["kernel"][:12], d["roofline"]["ms_per_step"], d["roofline"]["frac"], "| 2nd", d["roofline_gemm2"]["kernel"][:12], d["roofline_gemm2"]["ms_per_step"], d["roofline_gemm2"]["frac"])
PY
done
