# usage: bash tools/experiments/small_ab.sh [variant ...]: synthesis of one short utterance with the default library and each variants/lib_<v>.so
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" "$@"; do
  echo "== lib=${v:-default}"
  WGFLOW_LIB=${v:+$R/variants/lib_$v.so} python tools/experiments/infer_profile.py 63 2>&1 | grep MHz
  WGFLOW_LIB=${v:+$R/variants/lib_$v.so} python tools/experiments/infer_profile.py 40 2>&1 | grep MHz
done
done
