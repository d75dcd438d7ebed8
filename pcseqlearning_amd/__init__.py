"""MI355X-native hot path of PCSeqLearning (see README.md / DESIGN.md)."""
