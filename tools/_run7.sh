bash tools/collect.sh probe 200
OUT=fuzz_plain bash tools/collect.sh fuzz 411 700
OUT=fuzz_seg bash tools/collect.sh fuzz 412 500 segments
OUT=fuzz_big bash tools/collect.sh fuzz 413 80 big
