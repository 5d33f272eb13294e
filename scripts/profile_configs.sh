# rocprofv3 kernel statistics of the other configurations (rough box = C1b at bench scale, film = config 3, STL wire = config 4)
#   gpurun -- 'bash scripts/profile_configs.sh'  ->  gpurun_out/prof_cfg/*.csv
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ttrrp -o kt --output-format csv -- python3 $R/scripts/ttrrp_probe.py 1e7 31 > $O/ttrrp.log 2>&1
echo "ttrrp done"
rocprofv3 --kernel-trace --stats -d $O/c3 -o kt --output-format csv -- python3 $R/scripts/full_c3.py 1e7 500 > $O/c3.log 2>&1
echo "c3 done"
rocprofv3 --kernel-trace --stats -d $O/c4 -o kt --output-format csv -- python3 $R/scripts/full_c4.py 5e7 100 31 > $O/c4.log 2>&1
echo "c4 done"
for c in ttrrp c3 c4; do cp $O/$c/kt_kernel_stats.csv $O/${c}_kernel_stats.csv; head -4 $O/${c}_kernel_stats.csv | cut -c1-160; done
