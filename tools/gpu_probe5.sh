#!/bin/bash
timeout -k 5 60 ./tools/bin/fgprobe 4096 200 30 1
timeout -k 5 60 ./tools/bin/fgprobe 4096 200 30 257
timeout -k 5 60 ./tools/bin/fgprobe 1024 200 30 1 | head -3
