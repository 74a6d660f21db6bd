"""Extracts the serialized Inspector values of the reference's four Unity scenes into a JSON fixture.

    python tests/golden/make_unity_scene_values.py [/root/reference]  ->  tests/golden/unity_scene_values.json

Assets/Scenes/Scene{1..4}.unity are DATA files of the reference (Unity's text serialization of the scene objects), the
only place where the reference holds concrete inputs of the path: the SPH component's parameters (SPH.cs:5-49), its
spawn boxes (SpawnBox.cs:7-23 + the box's Transform position, SpawnBox.cs:25), the boundary cube's Transform
(SPH.cs:303-346 reads it), the obstacle's SphereCollider + Transform (SPH.cs:187-199) and the CircularMotion on it
(SphereMover.cs:6-10).  This script reads them as TEXT (a few regular expressions over the "--- !u!<class> &<fileID>"
documents; nothing is executed) and writes only the values.  tests/test_unity_scene_values.py compares
sph_amd.scenes.unity_scene with the fixture, so the scene builders are pinned by the reference's own data.

The reference is not present on the GPU box: the fixture is committed, this script is run by hand here.
"""
import json
import os
import re
import sys

# script GUIDs from Assets/<name>.cs.meta
GUID = {"8dd2fa90653633943ac9050adddb5336": "SPH", "6066f8eb70b39474792468eb6f155b5f": "SpawnBox",
        "6620c80153b13a4499c105429e18590a": "CircularMotion"}

_num = re.compile(r"^-?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$")


def _scalar(text):
    text = text.strip()
    if _num.match(text):
        return float(text) if any(ch in text for ch in ".eE") else int(text)
    return text


def _value(text):
    text = text.strip()
    if text.startswith("{") and text.endswith("}"):
        out = {}
        for part in text[1:-1].split(","):
            if ":" in part:
                k, v = part.split(":", 1)
                out[k.strip()] = _scalar(v)
        return out
    return _scalar(text)


def parse_documents(text):
    """fileID -> dict(kind=<first line, e.g. 'Transform'>, class_id, fields{top-level key: value or list})."""
    docs = {}
    for chunk in re.split(r"^--- ", text, flags=re.M)[1:]:
        lines = chunk.split("\n")
        m = re.match(r"!u!(\d+) &(\d+)", lines[0])
        if not m:
            continue
        fields, last_list = {}, None
        for ln in lines[2:]:
            if ln.startswith("  - ") and last_list is not None:
                fields[last_list].append(_value(ln[4:]))
                continue
            mm = re.match(r"^  (\w+):(.*)$", ln)
            if not mm:
                continue
            key, rest = mm.group(1), mm.group(2).strip()
            if rest == "":
                fields[key] = []
                last_list = key
            else:
                fields[key] = _value(rest)
                last_list = None
        docs[int(m.group(2))] = dict(kind=lines[1].rstrip(":"), class_id=int(m.group(1)), fields=fields)
    return docs


def _script(doc):
    s = doc["fields"].get("m_Script")
    return GUID.get(s.get("guid")) if isinstance(s, dict) else None


def _xyz(d, keys="xyz"):
    return [d[k] for k in keys]


def _transform_of(docs, game_object):
    for d in docs.values():
        if d["kind"] == "Transform" and d["fields"].get("m_GameObject", {}).get("fileID") == game_object:
            f = d["fields"]
            assert f["m_Father"]["fileID"] == 0, "a parented Transform: local values are not world values"
            return dict(position=_xyz(f["m_LocalPosition"]), rotation=_xyz(f["m_LocalRotation"], "xyzw"),
                        localScale=_xyz(f["m_LocalScale"]))
    raise KeyError(game_object)


def extract(path):
    docs = parse_documents(open(path, encoding="utf-8").read())
    sph = [d for d in docs.values() if d["kind"] == "MonoBehaviour" and _script(d) == "SPH"]
    assert len(sph) == 1, "one SPH component per scene"
    f = sph[0]["fields"]
    skip = {"m_ObjectHideFlags", "m_CorrespondingSourceObject", "m_PrefabInstance", "m_PrefabAsset", "m_GameObject",
            "m_Enabled", "m_EditorHideFlags", "m_Script", "m_Name", "m_EditorClassIdentifier", "fluidMaterial",
            "sphCompute", "spawnBoxes", "boundaryCube", "obstacleCollider"}
    out = {"sph": {k: (_xyz(v) if isinstance(v, dict) else v) for k, v in f.items() if k not in skip}}
    cube_id = f["boundaryCube"]["fileID"]
    out["boundaryCube"] = None
    if cube_id:
        out["boundaryCube"] = _transform_of(docs, docs[cube_id]["fields"]["m_GameObject"]["fileID"])
    out["obstacle"] = None
    coll_id = f["obstacleCollider"]["fileID"]
    if coll_id:
        coll = docs[coll_id]
        assert coll["kind"] == "SphereCollider"
        go = coll["fields"]["m_GameObject"]["fileID"]
        ob = dict(radius=coll["fields"]["m_Radius"], center=_xyz(coll["fields"]["m_Center"]),
                  transform=_transform_of(docs, go), mover=None)
        for d in docs.values():
            if d["kind"] == "MonoBehaviour" and _script(d) == "CircularMotion" and d["fields"]["m_GameObject"]["fileID"] == go:
                m = d["fields"]
                ob["mover"] = dict(enabled=m["m_Enabled"], center=_xyz(m["center"]), radius=m["radius"], speed=m["speed"])
        out["obstacle"] = ob
    boxes = []
    for ref in f["spawnBoxes"]:
        b = docs[ref["fileID"]]
        assert _script(b) == "SpawnBox"
        bf = b["fields"]
        go = bf["m_GameObject"]["fileID"]
        boxes.append(dict(particleCount=bf["particleCount"], spawnSize=_xyz(bf["spawnSize"]), restDensity=bf["restDensity"],
                          viscosity=bf["viscosity"], particleMass=bf["particleMass"],
                          initialVelocity=_xyz(bf["initialVelocity"]), particleColor=_xyz(bf["particleColor"], "rgba"),
                          position=_transform_of(docs, go)["position"], enabled=bf["m_Enabled"],
                          active=docs[go]["fields"]["m_IsActive"]))
    out["spawnBoxes"] = boxes
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    scenes = {}
    for k in (1, 2, 3, 4):
        scenes[f"Scene{k}"] = extract(os.path.join(ref, "Assets", "Scenes", f"Scene{k}.unity"))
    with open(os.path.join(here, "unity_scene_values.json"), "w") as fh:
        json.dump({"source": "Assets/Scenes/Scene{1..4}.unity of CPearson109/SPH (serialized Inspector values only)",
                   "scenes": scenes}, fh, indent=1, sort_keys=True)
        fh.write("\n")
    for name, sc in scenes.items():
        print(name, len(sc["spawnBoxes"]), "spawn boxes,", sum(b["particleCount"] for b in sc["spawnBoxes"]), "particles")


if __name__ == "__main__":
    main()
