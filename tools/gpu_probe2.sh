#!/bin/bash
W=./tools/bin/wrbench
for S in 1 4 16 52; do timeout -k 5 60 $W 0 $S 0; done
for S in 1 4 16 52; do timeout -k 5 60 $W 0 $S 22016; done
for G in 4 16 64 256; do timeout -k 5 60 $W 1 52 22016 $G; done
for D in 1 4 16; do timeout -k 5 60 $W 2 52 22016 8 $D; done
for D in 4 16; do timeout -k 5 60 $W 2 52 0 8 $D; done
