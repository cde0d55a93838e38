"""Host-side Python helpers of the MI355X-native ICP hot path (ctypes binding, workloads, mesh I/O)."""
