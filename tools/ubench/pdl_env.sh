for env in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=2"; do
  echo "=== env: $env"
  env $env timeout 100 ./tools/ubench/pdl_probe 40 2>&1 | grep -v "graph stage [3-7]"
done
