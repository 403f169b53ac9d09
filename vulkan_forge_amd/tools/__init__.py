"""Command-line tools over the drop-in API.  Same flags and report keys as the reference's
python/tools/{perf_sanity,determinism_harness,device_diagnostics,terrain_spike}.py so that existing CI invocations keep
working; each adds a terrain workload next to the reference's triangle one.  Run as `python -m vulkan_forge_amd.tools.<name>`."""
