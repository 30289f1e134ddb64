"""Track head building blocks on HIP kernels (reference iggt/heads/track_modules/)."""
