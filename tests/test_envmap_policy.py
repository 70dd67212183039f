"""Host mirror of the engine's HDRI resolution policy (Source/Engine/EnvironmentMap.cpp:69-136), SURVEY.md §8(f).3."""
from vqengine_amd import scene


def test_resolution_follows_the_monitor_height():
    p = "Data/Textures/HDRI/venice_sunset_%resolution%.hdr"
    for h, want in ((600, "1k"), (719, "1k"), (720, "2k"), (1079, "2k"), (1080, "4k"), (1440, "4k"), (1441, "8k"), (2159, "8k"), (2160, "8k"), (4320, "8k")):
        res, path = scene.determine_resolution_hdri(p, h)
        assert res == want and path == f"Data/Textures/HDRI/venice_sunset_{want}.hdr", (h, res, path)
    assert scene.determine_resolution_hdri("Data/Textures/HDRI/fixed_2k.hdr", 2160) == ("1k", "Data/Textures/HDRI/fixed_2k.hdr")   # no token: untouched


def test_downsize_source_lookup():
    files = ["HDRI/readme.txt", "HDRI/venice_sunset_preview.png", "HDRI/other_map_8k.hdr", "HDRI/venice_sunset_8k.hdr", "HDRI/venice_sunset_4k.hdr"]
    assert scene.find_environment_map_to_downsize_from(files, "venice_sunset", "4k") == "HDRI/venice_sunset_8k.hdr"     # first valid match wins
    assert scene.find_environment_map_to_downsize_from(files, "venice_sunset", "1k") == "HDRI/venice_sunset_8k.hdr"
    assert scene.find_environment_map_to_downsize_from(files, "venice_sunset", "8k") == ""                               # nothing above 8k
    assert scene.find_environment_map_to_downsize_from(files, "missing_map", "2k") == ""
    assert scene.HDRI_DIMENSIONS[4] == (4096, 2048) and scene.HDRI_DIMENSIONS[1] == (1024, 512)
