export TMPDIR=/tmp
for n in 2e5 1.25e7; do
rm -rf /tmp/tg && rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python3 tools/step_only.py $n 1024 400 > /tmp/tg.log 2>&1 && echo "n=$n" && python3 tools/trace_gaps.py /tmp/tg/t_kernel_trace.csv 600
done
