export LP_NUM_THREADS=32 OMP_NUM_THREADS=32
mkdir -p gpurun_out/r06_z2
timeout 1200 python tools/fuzz_variants_vs_reference_gl.py --device --n 120 --seed 18 --verbose > gpurun_out/r06_z2/variants_seed18_verbose.txt 2>&1
grep -c UNEXPLAINED gpurun_out/r06_z2/variants_seed18_verbose.txt
timeout 600 python tools/fuzz_variants_vs_reference_gl.py --device --n 40 --seed 19 --only-envmis --verbose > gpurun_out/r06_z2/envmis_seed19_verbose.txt 2>&1
tail -1 gpurun_out/r06_z2/envmis_seed19_verbose.txt | cut -c1-300
