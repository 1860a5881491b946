set -e
mkdir -p gpurun_out
out=gpurun_out/plan_ab.log
: > $out
for rep in 1 2; do
for wl in text source binary mixed; do
for f in 1 0; do
  echo "fused=$f $wl 1GiB" >> $out
  SFH_PLAN_FUSED=$f SF_WORKLOAD=$wl timeout -k 10 200 python tools/k1_time.py 1073741824 >> $out 2>&1
done; done; done
for wl in text source binary mixed; do
for f in 1 0; do
  echo "fused=$f $wl 256MiB" >> $out
  SFH_PLAN_FUSED=$f SF_WORKLOAD=$wl timeout -k 10 200 python tools/k1_time.py 268435456 >> $out 2>&1
done; done
grep -E "fused|k_plan" $out | paste - - | sed -E 's/.*(fused=[01] [a-z]+ [0-9A-Za-z]+).*k_plan.: ([0-9.]+).*/\1 \2/'
