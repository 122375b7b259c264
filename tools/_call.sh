set -u
mkdir -p gpurun_out/r03b
export FIESTA_ENVELOPE_LOG=$PWD/gpurun_out/r03b/envelope2.jsonl
rm -f $FIESTA_ENVELOPE_LOG
( time python -m pytest tests/test_gpu_dense_parity.py -m gpu -q --timeout 900 -k "sliding or window_then" 2>&1 | grep -v new_size | tail -30 ) 2>&1
