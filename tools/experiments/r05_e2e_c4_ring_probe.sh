#!/bin/bash
# ON THE GPU BOX: config-4-sized cohort end to end under three settings of the ring buffers
mkdir -p gpurun_out/parser
for tag in default pin_none chunk64; do
  case $tag in
    default) E="";;
    pin_none) E="AMPLISOLVE_PIN=none";;
    chunk64) E="AMPLISOLVE_CHUNK_MB=64";;
  esac
  env $E timeout -k 10 300 python bench.py --e2e c4 > gpurun_out/parser/e2e_c4_$tag.json 2> gpurun_out/parser/e2e_c4_$tag.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/parser/e2e_c4_$tag.json").read().strip().splitlines()[-1])["e2e"]["c4"]
for k in ("error_estimation","variant_calling"):
    e=d[k]; b=e["breakdown_s"]; print("$tag", k, round(e["wall_s"],3), "in_main", b["wall_in_main"], "outside", e.get("outside_main_s"), e.get("outside_main"), "parser_busy", b["parser_busy*"], "wait_parser", b["wait_for_parser"], "wait_ctx", b["wait_for_context"], "dev_wait", b["device_wait"], "reg", b.get("host_register"))
PY
done
