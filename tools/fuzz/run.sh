#!/bin/bash
# AddressSanitizer + UBSan fuzz of the host-side record readers and tokenizer (csrc/records.cpp, csrc/wordpiece.cpp) on the CPU build: damaged LMDB files (byte flips in page
# headers / node tables, truncation), damaged msgpack datapoints (exact-size heap copies, so any over-read is caught), random base64 text.
#   bash tools/fuzz/run.sh [iterations=3000]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d /tmp/fuzz_records.XXXX)
cat > $W/util.h <<'H'
#pragma once
namespace vk { int set_error(const char* fmt, ...); }
H
cp $R/volta_amd/csrc/records.cpp $W/records_copy.cpp       # compiled next to the stub util.h (the real one pulls in the HIP runtime)
cp $R/volta_amd/csrc/wordpiece.cpp $W/wordpiece_copy.cpp; cp $R/volta_amd/csrc/unicode_tables.h $W/
R=$R python3 - "$W" <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import numpy as np
from tests.lmdb_writer import pack_datapoint, write_lmdb
from tests.test_readers_cpu import _datapoint, _records
w = sys.argv[1]
write_lmdb(w + "/a.lmdb", _records(60), max_keys=4)
open(w + "/rec.bin", "wb").write(pack_datapoint(_datapoint(np.random.default_rng(0), 5)))
open(w + "/vocab.txt", "w", encoding="utf-8").write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "dog", "cat", "a", "b", "c", "##a", "##b", "##c", ".", ",", "!", "\u00e9", "e", "\u4e2d"]) + "\n")
PY
sed -i "s#/tmp/asan/m.lmdb#$W/m.lmdb#g" $W/records_copy.cpp
sed "s#/tmp/asan/m.lmdb#$W/m.lmdb#g" $R/tools/fuzz/fuzz_records.cpp > $W/driver.cpp
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$W -I$R/include -pthread $W/driver.cpp $W/records_copy.cpp $W/wordpiece_copy.cpp -o $W/fuzz
$W/fuzz $W/a.lmdb $W/rec.bin ${1:-3000} $W/vocab.txt
rm -rf $W
