for nb in 64 128 192 256 384 512 768; do python tools/kbench.py conv6430p 30 $nb 2>&1 | tail -1; done
