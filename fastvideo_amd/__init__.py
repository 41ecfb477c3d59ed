"""fastvideo_amd — MI355X (gfx950 / CDNA4) native kernels + host for FastVideo's per-step Wan video-DiT path.
See DESIGN.md / INTEGRATION.md.  Importing this package never touches the GPU."""
__version__ = "0.1.0"
