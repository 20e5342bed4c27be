#!/bin/bash
# The four tile shapes of the chained launch on representative ResNet50 layers at 512 images, each as a launch
# of its own: in-kernel stamps (diagnostic build, tools/libgdpso_stamps.so) and the plain launch time.
set -e
for shape in "512 1 2048 512 1 1 0 0 0 0" "512 2 1024 256 1 1 0 0 0 0" "512 4 512 128 1 1 0 0 0 0" "512 2 256 1024 1 1 0 0 0 1" "512 4 128 128 3 1 1 0 0 0"; do
  echo "== $shape"
  timeout -k 5 60 python tools/stamp_probe.py $shape
  timeout -k 5 60 python tools/probe_one.py $shape 200
done
