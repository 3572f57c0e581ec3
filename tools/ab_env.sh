# Run ON THE GPU BOX: the same python command alternately under two settings of one environment variable.
#   bash tools/ab_env.sh VAR A B ROUNDS cmd...
VAR=$1; A=$2; B=$3; N=$4; shift 4
for i in $(seq 1 $N); do
  for v in $A $B; do echo "== $VAR=$v"; env $VAR=$v "$@" 2>/dev/null | grep -v amdgpu; done
done
