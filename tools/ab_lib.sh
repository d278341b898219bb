#!/bin/bash
# A/B of two builds inside one GPU call: tools/ab_lib.sh <other.so> <command...>  runs the command with the in-tree library ("new"),
# then with <other.so> copied over it ("old"), and restores the library.
other=$1; shift
lib=discrete_dppm_graphs_amd/libdppm_hip.so
cp $lib /tmp/libdppm_new.so
TAG=new "$@" || exit 1
cp "$other" $lib
TAG=old "$@"; rc=$?
cp /tmp/libdppm_new.so $lib
exit $rc
