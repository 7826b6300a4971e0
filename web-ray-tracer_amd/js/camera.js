'use strict';
// Camera of the FlexLight API (reference modules/camera.js:3-11): plain data read by the renderer every frame.
class Camera {
  constructor () {
    this.x = 0; this.y = 0; this.z = 0;      // position
    this.fx = 0; this.fy = 0;                // yaw / pitch in radians
    this.fov = 1 / Math.PI;
  }
}
module.exports = { Camera };
