#!/bin/bash
# Same contract as the reference's run.sh (run.sh:1-7): `./run.sh fluid` (or `./run.sh mpm`) builds the
# program named by $1 from the sources in this repository and runs it with no arguments.
set -e
cd "$(dirname "$0")"
PKG="fluid-simulation_amd"
if [ "$1" != "fluid" ] && [ "$1" != "mpm" ]; then
    echo "usage: ./run.sh fluid | ./run.sh mpm" >&2
    exit 2
fi
make -C "$PKG" -s all
"./$PKG/$1"
