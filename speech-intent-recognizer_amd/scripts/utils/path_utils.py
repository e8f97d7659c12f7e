"""Audio path resolution for the CSV preprocessing step (reference: scripts/utils/path_utils.py:11-37)."""
import os


def normalize_audio_path(path, base_path):
    """Absolute paths pass through; relative ones are tried as given, under ``base_path`` and under the
    FSC dataset folders the reference probes; the original string comes back when nothing exists."""
    if os.path.isabs(path):
        return path
    fsc = os.path.join(base_path, "data", "FSC", "fluent_speech_commands_dataset")
    for loc in (path, os.path.join(base_path, path), os.path.join(fsc, path), os.path.join(fsc, "wavs", path)):
        if os.path.exists(loc):
            return loc
    print(f"Warning: Could not find audio file at {path}")
    return path
