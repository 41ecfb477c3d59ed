#!/bin/bash
# what kind of box is this? (boxes differ by up to 10 % in sustained MFMA throughput)
hostname 2>/dev/null
rocm-smi --showmaxpower --showpower --showtemp --showclocks --showperflevel --showserial 2>/dev/null | grep -v "^=\|^$" | cut -c1-120 | head -30
