set -e
for shape in "256 2 1024 256 1 1 0 0 0 0" ; do
for b in "" _skip1 _skip2 _skip4 _skip6; do
  echo "== $shape lib$b"
  GDPSO_STAMPS_LIB=tools/libgdpso_stamps$b.so timeout -k 5 60 python tools/stamp_probe.py $shape 2>/dev/null | grep -E "launch|K loop|prologue \(|epilogue"
done; done
