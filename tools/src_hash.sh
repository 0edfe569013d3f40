#!/bin/bash
# The content hash csrc/Makefile compiles into libnbmf_hip.so (nbmf_source_hash), computed from the tree: equal hashes =
# the library was built from these sources.  usage: tools/src_hash.sh   (from anywhere inside the repository)
cd "$(dirname "$0")/../nbmf_mm_amd/csrc" || exit 1
cat nbmf_hip.hip $(ls *.inc | LC_ALL=C sort) ../../include/nbmf_hip.h | sha256sum | cut -c1-12
