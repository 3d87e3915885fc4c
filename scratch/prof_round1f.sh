set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r1f_f16x2 -o s -- python bench.py --steps 100 --warmup 20 --no-cpu > $O/r1f_f16x2.json 2> $O/r1f_f16x2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r1f_f32 -o s -- python bench.py --steps 100 --warmup 20 --no-cpu --mode f32 > $O/r1f_f32.json 2> $O/r1f_f32.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r1f_f16x2_fetch -o f -- python bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r1f_f16x2_write -o f -- python bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/r1f_f16x2_sq -o f -- python bench.py --steps 6 --warmup 2 --no-cpu > /dev/null 2>&1
python profiles/summarize_pmc.py $O/r1f_pmc_summary.json fetch=$(ls $O/r1f_f16x2_fetch/*counter_collection.csv) write=$(ls $O/r1f_f16x2_write/*counter_collection.csv) sq=$(ls $O/r1f_f16x2_sq/*counter_collection.csv)
find $O/r1f_f16x2_fetch $O/r1f_f16x2_write $O/r1f_f16x2_sq -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -path "*r1f*" -delete
tail -1 $O/r1f_f16x2.json | cut -c1-300
python bench.py > $O/r1f_bench_default.json 2> $O/r1f_bench_default.err
python bench.py --mode f32 --no-cpu > $O/r1f_bench_f32.json 2>/dev/null
python bench.py --config cfg2 --mode f32 --steps 200 --warmup 20 --no-cpu > $O/r1f_bench_cfg2.json 2>/dev/null
python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu > $O/r1f_bench_cfg5.json 2>/dev/null
tail -c 600 $O/r1f_bench_default.json
