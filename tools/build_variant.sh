#!/bin/bash
# A measurement build of the library beside the product one:  tools/build_variant.sh NAME "-DDS_...=1 ..."  -> diffsci_amd/_lib_NAME/libdiffsci_hip.so
# (git-ignored like _lib/, travels to the GPU box; select it with DIFFSCI_HIP_LIB=diffsci_amd/_lib_NAME/libdiffsci_hip.so)
set -e
N=$1; F=$2
cd "$(dirname "$0")/.."
python - "$N" "$F" <<'PY'
import os, sys
import build
build.OUTDIR = os.path.join(build.ROOT, "diffsci_amd", "_lib_" + sys.argv[1])
build.LIB = os.path.join(build.OUTDIR, "libdiffsci_hip.so")
build.FLAGS = build.FLAGS + sys.argv[2].split()
print(build.build(force=True, verbose=False))
PY
