set -x
O=gpurun_out/r06_last; mkdir -p $O
export LP_NUM_THREADS=32 OMP_NUM_THREADS=32
timeout 1500 python tools/fuzz_variants_vs_reference_gl.py --device --n 900 --seed 203 > $O/fuzz_variants_device_vs_reference_gl_900_seed203.txt 2>&1; tail -1 $O/fuzz_variants_device_vs_reference_gl_900_seed203.txt | cut -c1-300
unset OMP_NUM_THREADS LP_NUM_THREADS
( time timeout 900 python -m pytest tests -q -m "gpu and quick" ) > $O/pytest_quick.log 2>&1; tail -5 $O/pytest_quick.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
