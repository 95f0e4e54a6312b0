bash tools/collect.sh suite || exit 1
bash tools/config2_probe.sh
bash tools/collect.sh probe 200
