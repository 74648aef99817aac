#!/bin/bash
# disassembly of one kernel of a built object: bash scratch/isa_dump.sh <obj> <demangled-name regex> > out.s
B=/opt/rocm/lib/llvm/bin
D=$(mktemp -d); cp $1 $D/x.o; (cd $D && $B/llvm-objdump --offloading x.o >/dev/null)
CO=$(ls $D/*gfx950* | head -1)
$B/llvm-objdump -d --demangle $CO | awk -v pat="$2" '/^[0-9a-f]+ <.*>:$/ {on = ($0 ~ pat)} on {print}'
rm -rf $D
