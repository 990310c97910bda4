"""hbetune_rs_amd: MI355X-native Gaussian-process surrogate engine for hbetune's src/gpr hot path."""
