cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_w.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc_t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_t.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_g -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_g.err
