'use strict';
// Config of the FlexLight API (reference modules/config.js:3-16), same field names; the renderer re-reads it
// every frame.  Defaults differ in two places so that a single renderFrame() is the plain path-traced frame:
// `temporal` and `antialiasing` default to off (the reference's defaults are true / 'fxaa').
// renderQuality scales the resolution the frame is traced at, as the reference sizes its canvas' drawing
// buffer (pathtracerWGL2.js:810-811).  firstPasses / secondPasses are forced to 3 by the reference's
// resize (pathtracerWGL2.js:818-819) and the pass schedule here is that fixed one.
class Config {
  constructor () {
    this.samplesPerRay = 1;
    this.renderQuality = 1;
    this.maxReflections = 5;
    this.minImportancy = 0.3;
    this.firstPasses = 3;
    this.secondPasses = 3;
    this.temporal = false;
    this.temporalSamples = 4;
    this.filter = false;
    this.hdr = true;
    this.antialiasing = undefined;
  }
}
module.exports = { Config };
