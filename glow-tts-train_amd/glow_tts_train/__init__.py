"""MI355X-native drop-in for the hot path of rhasspy/glow-tts-train: same module names and call surface
(`models.FlowGenerator`, `layers`, `attentions`, `utils`, `monotonic_align`, `optimize`), hand-written HIP underneath."""
