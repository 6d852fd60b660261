#!/bin/bash
# Dev tool: rebuild an object file from a hand-edited DEVICE assembly file.
#   asm_replay.sh DIR "<hipcc flags>" SRC.hip     (first run: full -save-temps build in DIR)
#   edit DIR/<name>-hip-amdgcn-amd-amdhsa-gfx950.s, then:  asm_replay.sh DIR --replay   -> DIR/out.o
set -e
DIR=$1; shift
mkdir -p $DIR; cd $DIR
if [ "$1" != "--replay" ]; then
  FLAGS=$1; SRC=$2
  /opt/rocm/bin/hipcc -### $FLAGS -save-temps -c $SRC -o out.o 2>&1 | grep '^ "' > cmds.txt
  /opt/rocm/bin/hipcc $FLAGS -save-temps -c $SRC -o out.o 2>/dev/null
  exit 0
fi
# commands: 1-3 device (E, bc, S), 4 device as, 5 lld, 6 bundler, 7 host E, 8 host bc, 9 host S, 10 host as
for i in 4 5 6 8 9 10; do
  sed -n "${i}p" cmds.txt > cmd.sh
  bash cmd.sh
done
