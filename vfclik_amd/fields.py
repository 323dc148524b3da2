"""Field-set maintenance of the vector-field executor, batched.

Each reference ``vf`` process keeps ``vectorFields = {id: [force, type, params]}`` and edits it from
``/param`` bottles (/root/reference/scripts/vf:145,209-275); on every message it rebuilds the summed
field (vf:276-293).  Here one :class:`FieldSets` keeps those dictionaries for B arms, applies the same
acceptance rules to each bottle, and hands the arms whose set changed to ``Engine.set_fields`` -- the
batched equivalent of the rebuild.  Malformed bottles are reported and ignored, never raised, like the
reference (vf:264-266,275).
"""
import logging

import numpy as np

from . import _abi

log = logging.getLogger("vfclik_amd.fields")

#: primitive types of the library (keys of vfl.vfl.vectorFieldLibrary() the reference uses: vf:148,238)
KNOWN_TYPES = (0, 1, 2, 4, 5)


class FieldSets:
    def __init__(self, batch, max_fields=16):
        self.batch = int(batch)
        self.max_fields = int(max_fields)
        self.sets = [dict() for _ in range(self.batch)]  # id -> [force, type, params]
        self.dirty = set()

    # -- one /param bottle for one arm (vf:212-275) ---------------------------------------------
    def handle_param(self, arm, bottle):
        """Apply an ``add`` / ``remove`` bottle.  Returns True when the arm's set changed."""
        if bottle is None or bottle.size() < 1:
            return False
        action = bottle.get(0).toString()
        vf = self.sets[arm]
        if action == "add":
            if bottle.size() != 5:  # vf:227,265-266
                log.warning("arm %d: wrong number of values, expected 5, ignoring", arm)
                return False
            vf_id = bottle.get(1).asInt()
            force = bottle.get(2).asDouble()
            vf_type = bottle.get(3).asInt()
            if vf_type not in KNOWN_TYPES:  # vf:238,263-264
                log.warning("arm %d: unknown vector field type %d, ignoring", arm, vf_type)
                return False
            plist = bottle.get(4).asList()
            params = [plist.get(i).asDouble() for i in range(plist.size())] if plist is not None else []
            need = _abi.FIELD_NPARAMS[vf_type]
            if len(params) < need:
                # the reference would fail later inside vfl's setParams; the batched library refuses here
                log.warning("arm %d: type %d needs %d parameters, got %d, ignoring", arm, vf_type, need, len(params))
                return False
            if vf_id not in vf and len(vf) >= self.max_fields:
                log.warning("arm %d: field capacity %d reached, ignoring id %d", arm, self.max_fields, vf_id)
                return False
            vf[vf_id] = [force, vf_type, params[:need]]
        elif action == "remove":
            if bottle.size() != 2:  # vf:268,274-275
                log.warning("arm %d: wrong number of values, expected 2", arm)
                return False
            vf_id = bottle.get(1).asInt()
            if vf_id not in vf:  # vf:269-273: silently nothing
                return False
            del vf[vf_id]
        else:
            return False
        self.dirty.add(arm)
        return True

    # -- direct (array) interface -------------------------------------------------------------------
    def set_arm(self, arm, fields):
        """fields: {id: [force, type, params]} replacing the arm's whole set."""
        for vf_id, (force, vf_type, params) in fields.items():
            if vf_type not in KNOWN_TYPES or len(params) < _abi.FIELD_NPARAMS[vf_type]:
                raise ValueError("field %d: bad type / parameter count" % vf_id)
        if len(fields) > self.max_fields:
            raise ValueError("more than %d fields" % self.max_fields)
        self.sets[arm] = {int(k): [float(v[0]), int(v[1]), [float(x) for x in v[2]]] for k, v in fields.items()}
        self.dirty.add(arm)

    def records(self, arms):
        """Structured array (len(arms), max_fields) + counts, ascending id, for Engine.set_fields."""
        arms = list(arms)
        rec = np.zeros((len(arms), self.max_fields), dtype=_abi.FIELD_DTYPE)
        cnt = np.zeros(len(arms), dtype=np.int32)
        for j, a in enumerate(arms):
            for k, vf_id in enumerate(sorted(self.sets[a])):
                force, vf_type, params = self.sets[a][vf_id]
                r = rec[j, k]
                r["id"], r["type"], r["force"] = vf_id, vf_type, force
                r["p"][: len(params)] = params
            cnt[j] = len(self.sets[a])
        return rec, cnt

    def flush(self, engine):
        """Upload the sets of all arms that changed since the last flush, as contiguous arm ranges."""
        if not self.dirty:
            return 0
        arms = sorted(self.dirty)
        self.dirty.clear()
        start = prev = arms[0]
        ranges = []
        for a in arms[1:]:
            if a != prev + 1:
                ranges.append((start, prev))
                start = a
            prev = a
        ranges.append((start, prev))
        for lo, hi in ranges:
            rec, cnt = self.records(range(lo, hi + 1))
            engine.set_fields(rec, cnt, first_arm=lo)
        return len(arms)
