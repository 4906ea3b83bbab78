for nw in 2 3 4 5; do echo "NW=$nw"; AADFF_CONV_NW=$nw python tools/kbench.py --rounds 9 2>&1 | grep conv_stack; done
