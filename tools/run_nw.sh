for nw in 2 4 5; do echo "NW=$nw"; AADFF_CONV_NW=$nw python tools/kbench.py --rounds 7 2>&1 | grep conv_stack; done
python tools/kbench.py --rounds 5 2>&1 | grep conv_single
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k 'render_psf or conv or stack_fused' 2>&1 | tail -3
